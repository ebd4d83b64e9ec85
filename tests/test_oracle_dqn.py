"""The DQN restatement (oracle/dqn.c) against what pins it: PyTorch-autograd vectors for the loss/target math
(tests/golden/torch_golden_dqn.json), the replay-eviction rule already pinned by the reference's own tests
(oracle_replay), an independent Python re-statement of ReplayBuffer / sample_minibatch written from
src/agents/buffers/replay.rs:89-165 and src/torch/agents/dqn.rs:280-291, and the schedule formulas of
src/torch/agents/schedules.rs.  CPU only."""
import collections
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O

L = O.lib()
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "torch_golden_dqn.json")) as f:
    GOLD = json.load(f)

QS = O.MlpShape(5, 16, 2)
KEY = [11, 22, 33, 44, 55, 66, 77, 88]


def _sim(n=24, max_steps=17, capacity=40, minibatch=60, td=False, seed=3):
    sim = O.LaneSim(n, max_steps=max_steps, seed_env=5, seed_actor=8)
    return O.DqnSim(sim, QS, O.mlp_init(QS, seed), capacity, KEY, minibatch, gamma=np.float32(0.99), one_step_td=td)


# ------------------------------------------------------------------ loss / targets vs torch autograd
@pytest.mark.parametrize("name,rtol", [("dqn_f32", 2e-5), ("dqn_f64", 1e-11)])
def test_dqn_gradient_against_torch_autograd(name, rtol):
    c = GOLD[name]
    D, H, A = c["dims"]
    s = O.MlpShape(D, H, A)
    a = np.array(c["actions"], np.int64)
    want = np.array(c["grad0"])
    if name.endswith("f32"):
        p, x, t = (np.array(c[k], np.float32) for k in ("params0", "obs", "td_targets"))
        g, loss = np.zeros_like(p), C.c_float()
        L.oracle_dqn_grad_f32(s, O.f32p(p), O.f32p(x), O.i64p(a), O.f32p(t), len(a), O.f32p(g), C.byref(loss))
    else:
        p, x, t = (np.array(c[k], np.float64) for k in ("params0", "obs", "td_targets"))
        g, loss = np.zeros_like(p), C.c_double()
        L.oracle_dqn_grad_f64(s, O.f64p(p), O.f64p(x), O.i64p(a), O.f64p(t), len(a), O.f64p(g), C.byref(loss))
    assert abs(loss.value - c["losses"][0]) <= rtol * abs(c["losses"][0])
    assert np.max(np.abs(g - want)) <= rtol * np.max(np.abs(want))


def test_one_step_td_targets_against_torch():
    """r + gamma * amax(Q(next)), terminal successors masked to 0 (critics/mod.rs:116-148; dqn.rs:300-309)."""
    c = GOLD["dqn_f32"]
    D, H, A = c["dims"]
    s = O.MlpShape(D, H, A)
    p = np.array(c["params0"], np.float32)
    q = O.mlp_forward_batch(s, p, np.array(c["next_obs"], np.float32))
    vnext = np.where(np.array(c["terminal"], bool), np.float32(0), q.max(axis=1))
    t = np.array(c["rewards"], np.float32) + np.float32(c["gamma"]) * vnext
    assert np.max(np.abs(t - np.array(c["td_targets"]))) < 2e-6


def test_adam_loop_against_torch():
    """n_backward_steps (torch/agents/mod.rs:35-72) with the DQN loss: losses before each step, final parameters."""
    c = GOLD["dqn_f32"]
    D, H, A = c["dims"]
    s = O.MlpShape(D, H, A)
    p, x, t = (np.array(c[k], np.float32) for k in ("params0", "obs", "td_targets"))
    a = np.array(c["actions"], np.int64)
    cfg = O.AdamCfg()
    L.oracle_adam_cfg_default(C.byref(cfg))
    st = L.oracle_adam_new(len(p))
    g = np.zeros_like(p)
    for k in range(c["steps"]):
        loss = C.c_float()
        L.oracle_dqn_grad_f32(s, O.f32p(p), O.f32p(x), O.i64p(a), O.f32p(t), len(a), O.f32p(g), C.byref(loss))
        assert abs(loss.value - c["losses"][k]) <= 3e-5 * abs(c["losses"][k])
        L.oracle_adam_step_f32(st, C.byref(cfg), O.f32p(p), O.f32p(g))
    L.oracle_adam_free(st)
    assert np.max(np.abs(p - np.array(c["params_final"]))) < 2e-6


# ------------------------------------------------------------------ schedules (schedules.rs:35-68)
def test_exploration_and_collection_schedules():
    er = L.oracle_exploration_rate
    assert er(1, 1.0, 0.1, 1000, 0, 1) == 1.0
    assert er(1, 1.0, 0.1, 1000, 500, 1) == 0.5 * (0.1 - 1.0) + 1.0
    assert er(1, 1.0, 0.1, 1000, 1000, 1) == 1.0 * (0.1 - 1.0) + 1.0
    assert er(1, 1.0, 0.1, 1000, 10 ** 9, 1) == 1.0 * (0.1 - 1.0) + 1.0
    assert er(1, 1.0, 0.1, 1000, 500, 0) == 0.0  # ActorMode::Evaluation
    assert er(0, 0.25, 0.0, 0, 123, 1) == 0.25
    us = L.oracle_collection_update_size
    b = us(1, 1_000_000, 100_000, 0)
    assert (b.min_steps, b.slack_steps) == (1_000_000, 1000)
    b = us(1, 1_000_000, 100_000, 999_999)
    assert b.min_steps == 1_000_000
    b = us(1, 1_000_000, 100_000, 1_000_000)
    assert (b.min_steps, b.slack_steps) == (100_000, 1000)
    b = us(0, 300, 7, 10 ** 9)
    assert (b.min_steps, b.slack_steps) == (300, 5)


# ------------------------------------------------------------------ collection + replay bookkeeping
class PyReplay:
    """ReplayBuffer (replay.rs:11-115) re-stated with collections.deque: steps are absolute step numbers."""

    def __init__(self, capacity):
        self.cap, self.steps, self.ends, self.offset, self.total = capacity, collections.deque(), collections.deque(), 0, 0

    def write(self, tag, done):
        if len(self.steps) == self.cap:
            assert self.ends, "Full"
            end = self.ends.popleft()
            for _ in range(end - self.offset):
                self.steps.popleft()
            self.offset = end
        self.steps.append(tag)
        self.total += 1
        if done:
            self.ends.append(self.total)

    def episodes(self):
        out, start = [], self.offset
        for e in self.ends:
            out.append((start, e - start))
            start = e
        return out


def test_collection_bookkeeping_matches_python_replay():
    d = _sim()
    n = d.sim.n
    mirrors = [PyReplay(d.capacity) for _ in range(n)]
    written = [0] * n
    for period, (T, eps) in enumerate([(25, 1.0), (30, 0.5), (18, 0.0), (40, 0.3)]):
        pos0 = d.actor_pos().copy()
        flags, full = d.collect(T, eps)
        assert not full
        assert np.all(flags[T - 1] != O.CONTINUE)  # horizon rule: every lane closes its episode
        for i in range(n):
            for t in range(T):
                mirrors[i].write(written[i], flags[t, i] != O.CONTINUE)
                written[i] += 1
        pos1 = d.actor_pos()
        if eps == 1.0:
            # gen_bool(1.0) draws nothing; gen_range(0..2) draws u64s until one passes sample_single's zone test
            # (for a power-of-two range rand 0.8.5's conservative zone rejects half of them)
            assert np.all(pos1 - pos0 >= 2 * T) and np.all((pos1 - pos0) % 2 == 0)
            assert np.any(pos1 - pos0 > 2 * T)
        if eps == 0.0:
            assert np.all(pos1 - pos0 == 2 * T)  # gen_bool(0.0) still draws its u64
        for i in range(n):
            ns, ne, tot = d.lane_info(i)
            m = mirrors[i]
            assert (ns, ne, tot) == (len(m.steps), len(m.ends), m.total)
            tags, lens = d.lane_dump(i)
            assert list(tags) == list(m.steps)
            assert [int(x) for x in lens] == [ln for _, ln in m.episodes()]
            assert ns <= d.capacity
    # something was evicted, and whole episodes only
    assert any(m.offset > 0 for m in mirrors)


def test_greedy_actions_are_argmax_and_random_actions_are_gen_range():
    d = _sim(n=8, capacity=200)
    T = 30
    d.collect(T, 0.0)
    for i in range(d.sim.n):
        tags, _ = d.lane_dump(i)
        for k in tags:
            obs, a, r, nx, nobs = d.step_data(i, int(k))
            q = O.mlp_forward_batch(QS, d.qparams, obs[None, :])[0]
            assert a == (1 if q[1] > q[0] else 0)
            assert r == 1.0
    d2 = _sim(n=8, capacity=200)
    d2.collect(T, 1.0)
    for i in range(d2.sim.n):
        rng = O.Prng()
        L.oracle_prng_seed_from_u64(C.byref(rng), 8)
        L.oracle_prng_set_stream(C.byref(rng), i)
        L.oracle_prng_set_word_pos(C.byref(rng), 0)
        tags, _ = d2.lane_dump(i)
        for k in tags:
            while True:  # UniformInt::sample_single(0, 2): zone = (2 << 62) - 1
                v = L.oracle_prng_next_u64(C.byref(rng))
                if ((v << 1) & ((1 << 64) - 1)) <= (1 << 63) - 1:
                    break
            assert d2.step_data(i, int(k))[1] == (v >> 63)  # widening multiply by 2: the top bit
        assert d2.actor_pos()[i] == L.oracle_prng_word_pos(C.byref(rng))


def test_full_buffer_is_reported():
    d = _sim(n=4, max_steps=500, capacity=6)
    # a CartPole episode under the greedy policy of a fresh net lasts longer than 6 steps
    _, full = d.collect(30, 0.0)
    assert full


# ------------------------------------------------------------------ minibatch sampling
def _py_sample(d, rng, minibatch_steps):
    """dqn.rs:280-291 with Python ints; `rng` is an oracle Prng (u64 draws only)."""
    n = d.sim.n
    out, total, cand = [], 0, 0
    while True:
        lane = cand % n
        tags, lens = d.lane_dump(lane)
        ne = len(lens)
        zone = (1 << 64) - 1 - (((1 << 64) - ne) % ne)
        while True:
            v = L.oracle_prng_next_u64(C.byref(rng))
            m = v * ne
            if (m & ((1 << 64) - 1)) <= zone:
                idx = m >> 64
                break
        start_rel = int(sum(int(x) for x in lens[:idx]))
        take = total < minibatch_steps
        if not take:
            return out, total
        total += int(lens[idx])
        out.append((lane, int(tags[start_rel]), int(lens[idx])))
        cand += 1


def test_sampling_matches_python_restatement_and_consumes_one_extra_draw():
    d = _sim()
    d.collect(35, 0.7)
    d.collect(35, 0.7)
    rng = O.Prng()
    key = (C.c_uint32 * 8)(*KEY)
    L.oracle_prng_from_seed(C.byref(rng), key)
    for _ in range(3):
        lanes, starts, lens, ns = d.sample()
        want, total = _py_sample(d, rng, d.minibatch_steps)
        assert [(int(a), int(b), int(c)) for a, b, c in zip(lanes, starts, lens)] == want
        assert ns == total == int(lens.sum())
        assert total >= d.minibatch_steps and total - int(lens[-1]) < d.minibatch_steps
        assert d.agent_pos() == L.oracle_prng_word_pos(C.byref(rng))
    # lanes are visited round-robin from lane 0 in every minibatch
    assert list(lanes[: d.sim.n]) == list(range(min(d.sim.n, len(lanes))))


@pytest.mark.parametrize("td", [False, True])
def test_minibatch_targets(td):
    d = _sim(td=td)
    d.collect(35, 0.5)
    lanes, starts, lens, ns = d.sample()
    obs, actions, targets = d.minibatch(lanes, starts, lens)
    off = 0
    gamma = np.float32(0.99)
    for lane, s0, ln in zip(lanes, starts, lens):
        steps = [d.step_data(int(lane), int(s0) + i) for i in range(int(ln))]
        assert steps[-1][3] != O.CONTINUE and all(s[3] == O.CONTINUE for s in steps[:-1])
        for i, s in enumerate(steps):
            assert np.array_equal(obs[off + i], s[0]) and actions[off + i] == s[1]
        if td:
            for i, s in enumerate(steps):
                if s[3] == O.TERMINATE:
                    vnext = np.float32(0)
                else:
                    nxt = s[4] if s[3] == O.INTERRUPT else steps[i + 1][0]
                    vnext = O.mlp_forward_batch(QS, d.qparams, nxt[None, :])[0].max()
                assert targets[off + i] == np.float32(s[2]) + gamma * vnext
        else:
            # the reference's discounted_cumsum_from_end (pinned by its own fixtures) on this one sequence
            r = np.array([s[2] for s in steps], np.float32)
            bs = np.ones(len(r), np.uint64)
            L.oracle_discounted_cumsum_from_end_f32(O.f32p(r), len(r), gamma, O.u64p(bs), len(bs))
            assert np.array_equal(targets[off:off + int(ln)], r)
        off += int(ln)
    assert off == ns


def test_update_loop_reduces_loss_and_is_reproducible():
    a, b = _sim(), _sim()
    for d in (a, b):
        d.collect(35, 1.0)
    la, lb = a.update(8), b.update(8)
    assert np.array_equal(la, lb) and np.array_equal(a.qparams, b.qparams)
    assert a.agent_pos() == b.agent_pos() > 0
    assert np.all(np.isfinite(la))
