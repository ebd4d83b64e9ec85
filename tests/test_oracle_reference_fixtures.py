"""The oracle against every known-answer fixture the reference's own unit tests hold for the path
(tests/golden/reference_fixtures.json, transcribed data only).  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle as O

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_fixtures.json")))
SUCC = {"C": O.CONTINUE, "T": O.TERMINATE, "I": O.INTERRUPT}
L = O.lib()


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def pack(seqs, dtype=np.float32):
    lens = u64([len(s) for s in seqs])
    n = int(lens.sum())
    seq_i, off_i = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    L.oracle_packed_order(O.u64p(lens), len(seqs), O.u64p(seq_i), O.u64p(off_i))
    return np.array([seqs[int(s)][int(o)] for s, o in zip(seq_i, off_i)], dtype=dtype), lens


def batch_sizes(lens):
    out = np.zeros(int(max(lens)) + 1, np.uint64)
    k = L.oracle_packed_batch_sizes(O.u64p(u64(lens)), len(lens), O.u64p(out), len(out))
    return k, out[:max(k, 0)]


def test_batch_sizes():
    f = FIX["packed_batch_sizes"]
    k, bs = batch_sizes(f["lengths"])
    assert k == 4 and bs.tolist() == f["batch_sizes"]
    k, _ = batch_sizes(f["increasing"])
    assert k == -1  # PackingError::Increasing


def test_packing_order():
    f = FIX["packed_order"]
    packed, _ = pack(f["sequences"], np.int64)
    assert packed.tolist() == f["packed"]


def test_trim_start_and_end():
    f = FIX["packed_trim"]
    packed, lens = pack(f["sequences"])
    _, bs = batch_sizes(lens)
    for n, key in ((1, "trim_start_1"), (3, "trim_start_3")):
        # view_trim_start(n): drop the first n time slices
        skip = int(bs[:n].sum())
        exp, _ = pack(f[key])
        assert packed[skip:].tolist() == exp.tolist()
        nb = np.zeros(len(bs), np.uint64)
        k = L.oracle_packed_trim_batch_sizes(O.u64p(bs), len(bs), n, O.u64p(nb))
        _, ebs = batch_sizes([len(s) for s in f[key]])
        assert nb[:k].tolist() == ebs.tolist()
    for n, key in ((1, "trim_end_1"), (3, "trim_end_3")):
        exp, _ = pack(f[key])
        out = np.zeros(len(exp), np.float32)
        L.oracle_packed_trim_end_f32(O.f32p(packed), O.u64p(bs), len(bs), n, O.f32p(out))
        assert out.tolist() == exp.tolist()


def test_discounted_cumsum_from_end():
    f = FIX["discounted_cumsum"]
    packed, lens = pack(f["sequences"])
    _, bs = batch_sizes(lens)
    L.oracle_discounted_cumsum_from_end_f32(O.f32p(packed), len(packed), f["discount"], O.u64p(bs), len(bs))
    exp, _ = pack(f["expected"])
    assert np.allclose(packed, exp, atol=f["atol"], rtol=0)


def _buffer_from(steps, obs_dim=1):
    b = L.oracle_vecbuffer_new(obs_dim)
    for st in steps:
        obs = np.array([st[0]], np.float32)
        L.oracle_vecbuffer_write_step(b, O.f32p(obs), 0, 0.0, SUCC[st[1]], None)
    L.oracle_vecbuffer_end_experience(b)
    return b


def test_vec_buffer_finalisation():
    f = FIX["vec_buffer"]
    b = _buffer_from(f["steps"])
    assert b.contents.len == f["num_steps"] and b.contents.n_episode_ends == f["num_episodes"]
    ends = np.zeros(f["num_episodes"], np.uint64)
    L.oracle_vecbuffer_episode_ends(b, O.u64p(ends))
    assert ends.tolist() == f["episode_ends"]
    n = f["num_steps"]
    obs, nxt, nobs = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32)
    L.oracle_vecbuffer_steps(b, O.f32p(obs), None, None, O.u8p(nxt), O.f32p(nobs))
    for i, (o, s, no) in enumerate(f["final_steps"]):
        assert obs[i] == o and nxt[i] == SUCC[s]
        if no is not None:
            assert nobs[i] == no
    L.oracle_vecbuffer_free(b)


def test_replay_buffer_eviction():
    f = FIX["replay_buffer"]
    r = L.oracle_replay_new(f["capacity"])
    for w in f["writes"]:
        for tag, s in w["steps"]:
            assert L.oracle_replay_write_step(r, tag, 1 if s != "C" else 0) == 0
        L.oracle_replay_end_experience(r)
        assert L.oracle_replay_num_steps(r) == w["num_steps"]
        assert L.oracle_replay_num_episodes(r) == w["num_episodes"]
        tags = np.zeros(w["num_steps"], np.int32)
        lens = np.zeros(w["num_episodes"], np.uint64)
        L.oracle_replay_dump(r, O.i32p(tags), O.u64p(lens))
        assert tags.tolist() == w["tags"] and lens.tolist() == w["episode_lens"]
    L.oracle_replay_free(r)
    r = L.oracle_replay_new(f["capacity"])
    full = [L.oracle_replay_write_step(r, 0, 0) for _ in range(f["too_large"])]
    assert 1 in full and full.index(1) == f["capacity"]  # WriteExperienceError::Full
    L.oracle_replay_free(r)


def test_take_aligned_steps():
    f = FIX["take_aligned_steps"]
    done = np.array(f["episode_done"], np.uint8)
    for c in f["cases"]:
        assert L.oracle_take_aligned_count(O.u8p(done), len(done), c["min"], c["slack"]) == c["taken"]


def test_history_data_bound():
    f = FIX["history_data_bound"]
    for c in f["divide"]:
        r = L.oracle_bound_divide(O.Bound(c["min"], c["slack"]), c["n"])
        assert [r.min_steps, r.slack_steps] == c["out"]
    for c in f["default_slack"]:
        assert L.oracle_bound_with_default_slack(c["min"]).slack_steps == c["slack"]
    m = L.oracle_bound_max(O.Bound(3, 9), O.Bound(7, 2))
    assert (m.min_steps, m.slack_steps) == (7, 9)


def test_step_limit_wrapper():
    f = FIX["step_limit"]
    rem = C.c_uint64(f["max_steps"])
    got_rem, got_succ = [], []
    for _ in f["successors"]:
        got_rem.append(L.oracle_step_limit_remaining(rem.value, f["max_steps"]))
        got_succ.append(L.oracle_step_limit_apply(O.CONTINUE, C.byref(rem)))
    assert got_rem == f["remaining"]
    assert got_succ == [SUCC[s] for s in f["successors"]]
    assert rem.value == f["final_steps_remaining"]
    rem = C.c_uint64(5)  # an inner Terminate passes through untouched
    assert L.oracle_step_limit_apply(O.TERMINATE, C.byref(rem)) == O.TERMINATE


def test_index_features():
    f = FIX["index_features"]
    out = np.full(f["size"], 7.0, np.float32)
    L.oracle_index_features(f["index"], f["size"], O.f32p(out))
    assert out.tolist() == f["one_hot"]


def test_history_features_packing():
    f = FIX["history_features"]
    b = L.oracle_vecbuffer_new(1)
    for ep in f["episodes"]:
        for obs, act, rew, nxt in ep:
            o = np.array([obs], np.float32)
            if nxt.startswith("I"):
                no = np.array([float(nxt[1:])], np.float32)
                L.oracle_vecbuffer_write_step(b, O.f32p(o), act, rew, O.INTERRUPT, O.f32p(no))
            else:
                L.oracle_vecbuffer_write_step(b, O.f32p(o), act, rew, SUCC[nxt], None)
        # the fixture's first episode ends with Continue: mark the episode boundary like the test's Vec of
        # episodes does (LazyHistoryFeatures receives explicit episode slices)
        if ep[-1][3] == "C":
            bb = b.contents
            assert bb.n_episode_ends == 0
            ends = (C.c_uint64 * 1)(bb.len)
            # push an episode end without altering the step (no successor observation: invalid extended row)
            L.oracle_vecbuffer_write_step(b, O.f32p(np.zeros(1, np.float32)), 0, 0.0, O.TERMINATE, None)
            b.contents.len -= 1
            b.contents.episode_ends[0] = b.contents.len
    arr = (C.POINTER(O.VecBuffer) * 1)(b)
    feat = L.oracle_features_from_buffers(arr, 1)
    ft = feat.contents
    assert ft.n_steps == f["num_steps"] and ft.n_episodes == f["num_episodes"]
    n = ft.n_steps
    assert [ft.obs[i] for i in range(n)] == f["observation_features"]
    assert [ft.actions[i] for i in range(n)] == f["actions"]
    assert [ft.rewards[i] for i in range(n)] == f["rewards"]
    assert [ft.batch_sizes[i] for i in range(ft.n_batches)] == f["batch_sizes"]
    # extended structure: every episode one longer; invalid rows = ends that are not Interrupt
    assert ft.n_ext == n + ft.n_episodes
    inv = [ft.is_invalid[i] for i in range(ft.n_ext)]
    assert sum(inv) == 3  # episodes ending in Continue(dangling) / Terminate / Terminate; one Interrupt(true)
    L.oracle_features_free(feat)
    L.oracle_vecbuffer_free(b)


def _log_softmax(z):
    z = np.array([float(v) if v != "-inf" else -np.inf for v in z], np.float32)
    lp = np.zeros_like(z)
    L.oracle_log_softmax_f32(O.f32p(z), len(z), O.f32p(lp), 1)
    return lp


def test_categorical_log_probs():
    f = FIX["categorical_log_probs"]
    ln = math.log(math.exp(-1.0) + 1.0 + math.exp(1.0))
    expected = [0.0, -np.inf, -math.log(2.0), -np.inf, -1.0 - ln, -ln, 1.0 - ln, math.log(1.0 / 3.0)]
    for z, e, exp in zip(f["logits"], f["elements"], expected):
        got = _log_softmax(z)[e]
        assert (got == exp) if np.isinf(exp) else abs(got - exp) <= f["atol"]


def test_categorical_entropy_and_kl():
    f = FIX["categorical_entropy"]
    expected = [0.0, -math.log(0.5), -math.log(1.0 / 3.0),
                -0.1 * math.log(0.1) - 0.3 * math.log(0.3) - 0.6 * math.log(0.6)]
    for p, exp in zip(f["probs"], expected):
        with np.errstate(divide="ignore"):
            lp = _log_softmax([("-inf" if v == 0 else math.log(v)) for v in p])
        assert abs(L.oracle_categorical_entropy_f32(O.f32p(lp), len(lp), 1) - exp) <= f["atol"]
        assert abs(L.oracle_categorical_entropy_f32(O.f32p(lp), len(lp), 0) - exp) <= 2 * f["atol"]
    f = FIX["categorical_kl"]
    expected = [0.0, 0.2 * math.log(0.2 / 0.7) + 0.3 * math.log(0.3 / 0.2) + 0.5 * math.log(0.5 / 0.1),
                math.log(1.0 / 0.3)]
    for pa, pb, exp in zip(f["probs_a"], f["probs_b"], expected):
        la = _log_softmax([("-inf" if v == 0 else math.log(v)) for v in pa])
        lb = _log_softmax([("-inf" if v == 0 else math.log(v)) for v in pb])
        assert abs(L.oracle_categorical_kl_f32(O.f32p(la), O.f32p(lb), 3, 1) - exp) <= f["atol"]


def test_categorical_sampling_extremes_and_frequencies():
    # deterministic rows always return their only outcome (categorical.rs:116-131)
    for k in range(3):
        p = [0.0, 0.0, 0.0]
        p[k] = 1.0
        lp = _log_softmax([("-inf" if v == 0 else 0.0) for v in p])
        for u in (0.0, 0.3, 0.999999):
            assert L.oracle_categorical_sample_u(O.f32p(lp), 3, u, 0) == k
    # frequencies within a 4.4 sigma Wald interval (spaces/index.rs:379-418 style)
    lp = _log_softmax([math.log(0.3), math.log(0.3), math.log(0.4)])
    r = O.Prng()
    L.oracle_prng_seed_from_u64(C.byref(r), 5)
    n = 20000
    counts = np.zeros(3)
    for _ in range(n):
        counts[L.oracle_categorical_sample_u(O.f32p(lp), 3, L.oracle_prng_gen_f32(C.byref(r)), 0)] += 1
    for k, p in enumerate((0.3, 0.3, 0.4)):
        assert abs(counts[k] / n - p) < 4.4 * math.sqrt(p * (1 - p) / n)


def test_cg_solve_2x2():
    f = FIX["cg_solve_2x2"]
    A, b = np.array(f["A"], np.float64), np.array(f["b"], np.float64)
    x = np.zeros(2, np.float64)
    L.oracle_cg_dense_f64(O.f64p(A), O.f64p(b), 2, f["iterations"], f["tol"], O.f64p(x))
    assert np.linalg.norm(x - np.array(f["x"])) < f["tol"]
    x32 = np.zeros(2, np.float32)
    L.oracle_cg_dense_f32(O.f32p(A.astype(np.float32)), O.f32p(b.astype(np.float32)), 2, 10, 1e-4, O.f32p(x32))
    assert np.linalg.norm(x32 - np.array(f["x"])) < 1e-3


def test_relu_through_the_mlp():
    f = FIX["relu"]
    s = O.MlpShape(1, 1, 1)
    params = np.array([1.0, 0.0, 1.0, 0.0], np.float32)  # W1, b1, W2, b2
    x = np.array(f["in"], np.float32).reshape(-1, 1)
    assert O.mlp_forward_batch(s, params, x)[:, 0].tolist() == f["out"]
