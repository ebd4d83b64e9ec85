"""The recurrent restatement (oracle/seq.c, seq_impl.inc) against what pins it: torch.gru_cell + autograd through
time on lane trajectories with episode boundaries (tests/golden/torch_golden_gru.json), the scalar Chain env and
step-limit rule already pinned by the reference's fixtures, and structural properties of the initialisation.
CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O

L = O.lib()
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "torch_golden_gru.json")) as f:
    GOLD = json.load(f)


def _case(name):
    c = GOLD[name]
    D, H, H2, A = c["dims"]
    n, T = c["n"], c["T"]
    s = O.GruShape(D, H, H2, A)
    traj = dict(obs=np.array(c["obs"]).reshape(D, T + 1, n), term_obs=np.array(c["term_obs"]).reshape(D, T, n),
                flag=np.array(c["flag"], np.uint8).reshape(T, n))
    return c, s, traj


@pytest.mark.parametrize("name,tol", [("gru_f32", 3e-6), ("gru_f64", 1e-13)])
def test_forward_outputs_and_successor_outputs(name, tol):
    c, s, traj = _case(name)
    f64 = name.endswith("f64")
    out, succ = O.gru_seq_forward(s, np.array(c["params"]), traj, f64=f64)
    assert np.max(np.abs(out.reshape(-1) - np.array(c["out"]))) < tol
    assert np.max(np.abs(succ.reshape(-1) - np.array(c["succ_out"]))) < tol
    assert np.count_nonzero(np.array(c["succ_out"])) > 0


@pytest.mark.parametrize("name,rtol", [("gru_f32", 2e-5), ("gru_f64", 1e-11)])
def test_backward_through_time_against_autograd(name, rtol):
    c, s, traj = _case(name)
    f64 = name.endswith("f64")
    A, T, n = s.out_dim, c["T"], c["n"]
    g = O.gru_seq_backward(s, np.array(c["params"]), traj, np.array(c["dout"]).reshape(A, T, n), f64=f64)
    want = np.array(c["grad"])
    assert np.max(np.abs(g - want)) <= rtol * np.max(np.abs(want))
    # every parameter block receives gradient (W_ih, W_hh, b_ih, b_hh, W1, b1, W2, b2)
    assert np.count_nonzero(want) > 0.9 * len(want)


def test_step_equals_sequence_forward():
    c, s, traj = _case("gru_f32")
    p = np.array(c["params"], np.float32)
    out, _ = O.gru_seq_forward(s, p, traj)
    obs = traj["obs"].astype(np.float32)
    for i in range(c["n"]):
        h = np.zeros(s.hidden, np.float32)
        for t in range(c["T"]):
            y = np.zeros(s.out_dim, np.float32)
            L.oracle_gru_step_f32(s, O.f32p(p), O.f32p(np.ascontiguousarray(obs[:, t, i])), O.f32p(h), O.f32p(y))
            assert np.array_equal(y, out[:, t, i])
            if traj["flag"][t, i] != 0:
                h[:] = 0


def test_initialisation_structure():
    s = O.GruShape(5, 128, 128, 2)
    p = O.gru_init(s, 9)
    H, D = 128, 5
    assert len(p) == 3 * H * D + 3 * H * H + 6 * H + H * H + H + 2 * H + 2 == 68610
    wih = p[:3 * H * D]
    assert np.abs(wih).max() <= np.float32(np.sqrt(6.0 / (D + 3 * H))) and np.abs(wih).max() > 0.11
    whh = p[3 * H * D:3 * H * D + 3 * H * H].reshape(3 * H, H).astype(np.float64)
    assert np.abs(whh.T @ whh - np.eye(H)).max() < 1e-6  # orthonormal columns (initializers.rs:383-393)
    b = p[3 * H * D + 3 * H * H:3 * H * D + 3 * H * H + 6 * H]
    assert not b.any()
    mlp = p[3 * H * D + 3 * H * H + 6 * H:]
    assert np.abs(mlp[:H * H + H]).max() <= np.float32(np.sqrt(6.0 / (H + 1 + H)))
    assert not np.array_equal(O.gru_init(s, 10), p)


def test_chain_lanes_follow_the_scalar_env():
    """lane i at global step t = Chain::step with the slip draw at word t of stream i; LatentStepLimit interrupts"""
    sim = O.ChainLaneSim(64, max_steps=7, seed_env=3, seed_actor=4)
    assert sim.D == 5
    env = O.Chain()
    L.oracle_chain_default(C.byref(env))
    rng = np.random.default_rng(0)
    state = np.zeros(64, np.uint64)
    rem = np.full(64, 7, np.uint64)
    for t in range(30):
        obs = sim.observe()
        assert np.array_equal(obs.argmax(0), state) and np.all(obs.sum(0) == 1)
        a = rng.integers(0, 2, 64).astype(np.uint8)
        reward, flag, nobs, term = sim.step(a)
        for i in range(64):
            r = O.Prng()
            L.oracle_prng_seed_from_u64(C.byref(r), 3)
            L.oracle_prng_set_stream(C.byref(r), i)
            L.oracle_prng_set_word_pos(C.byref(r), t)
            st = C.c_uint64(int(state[i]))
            rw = C.c_double()
            L.oracle_chain_step(C.byref(env), C.byref(st), int(a[i]), C.byref(r), C.byref(rw))
            rem[i] -= 1
            assert reward[i] == np.float32(rw.value)
            if rem[i] == 0:
                assert flag[i] == O.INTERRUPT and term[:, i].argmax() == st.value
                state[i], rem[i] = 0, 7
            else:
                assert flag[i] == O.CONTINUE
                state[i] = st.value
    assert set(np.unique(reward)) <= {0.0, 2.0, 10.0}


def test_rollout_teacher_forcing_consistency_and_gae():
    s = O.GruShape(5, 16, 12, 2)
    cs = O.GruShape(5, 16, 12, 1)
    sim = O.ChainLaneSim(40, max_steps=9, seed_env=5, seed_actor=6)
    p, cp = O.gru_init(s, 1), O.gru_init(cs, 2)
    traj = sim.rollout_gru(s, p, 25, threads=2)
    assert (traj["flag"] == O.INTERRUPT).sum() == 40 * (25 // 9)
    # the teacher-forced forward reproduces the behaviour policy: re-sampling with the same uniforms gives the
    # same actions
    logits, _ = O.gru_seq_forward(s, p, traj)
    for i in (0, 17, 39):
        r = O.Prng()
        L.oracle_prng_seed_from_u64(C.byref(r), 6)
        L.oracle_prng_set_stream(C.byref(r), i)
        L.oracle_prng_set_word_pos(C.byref(r), 0)
        for t in range(25):
            u = L.oracle_prng_gen_f32(C.byref(r))
            lp = np.zeros(2, np.float32)
            L.oracle_log_softmax_f32(O.f32p(np.ascontiguousarray(logits[:, t, i])), 2, O.f32p(lp), 0)
            assert L.oracle_categorical_sample_u(O.f32p(lp), 2, u, 0) == traj["action"][t, i]
    v, sv = O.gru_seq_forward(cs, cp, traj)
    adv, rtg = O.seq_gae(v[0], sv[0], traj, np.float32(0.95), np.float32(0.9))
    # reward-to-go restarts at every cut; the last step of each episode bootstraps from the successor value
    t_end = np.where(traj["flag"][:, 3] != 0)[0]
    assert np.array_equal(rtg[t_end, 3], traj["reward"][t_end, 3])
    te = t_end[0]
    delta = (traj["reward"][te, 3] + np.float32(0.95) * sv[0][te, 3]) - v[0][te, 3]
    assert adv[te, 3] == delta


@pytest.mark.parametrize("name,rtol", [("gru_f32", 3e-5), ("gru_f64", 1e-11)])
def test_forward_mode_derivative_through_time(name, rtol):
    """oracle_gru_seq_jvp against torch.autograd.functional.jvp of all step outputs along a parameter tangent"""
    c, s, traj = _case(name)
    f64 = name.endswith("f64")
    od = O.gru_seq_jvp(s, np.array(c["params"]), np.array(c["tangent"]), traj, f64=f64)
    want = np.array(c["out_dot"]).reshape(od.shape)
    assert np.max(np.abs(od - want)) <= rtol * np.max(np.abs(want))


def test_fisher_vector_product_is_symmetric_positive_and_matches_finite_differences():
    """F = J^T M J / B: v.Fw == w.Fv, v.Fv >= 0, and F v = d/de grad KL(pi_theta0 || pi_theta0+e v) at e -> 0"""
    s = O.GruShape(5, 16, 12, 2)
    sim = O.ChainLaneSim(24, max_steps=7, seed_env=2, seed_actor=3)
    p = O.gru_init(s, 4).astype(np.float64)
    traj = sim.rollout_gru(s, p.astype(np.float32), 18, threads=2)
    rng = np.random.default_rng(0)
    v, w = rng.normal(size=len(p)), rng.normal(size=len(p))
    Fv = O.gru_policy_fvp(s, p, v, traj, 0.0, f64=True)
    Fw = O.gru_policy_fvp(s, p, w, traj, 0.0, f64=True)
    assert abs(v @ Fw - w @ Fv) <= 1e-10 * abs(v @ Fw)
    assert v @ Fv > 0

    def kl_grad(theta):
        l0, _ = O.gru_seq_forward(s, p, traj, f64=True, want_succ=False)
        l1, _ = O.gru_seq_forward(s, theta, traj, f64=True, want_succ=False)
        lp0 = l0 - np.log(np.exp(l0).sum(0))
        lp1 = l1 - np.log(np.exp(l1).sum(0))
        # d KL(p0 || p1) / d logits1 = (p1 - p0) / B
        dz = (np.exp(lp1) - np.exp(lp0)) / l0[0].size
        return O.gru_seq_backward(s, theta, traj, dz, f64=True)

    e = 1e-5
    fd = (kl_grad(p + e * v) - kl_grad(p - e * v)) / (2 * e)
    assert np.max(np.abs(fd - Fv)) <= 2e-5 * np.max(np.abs(Fv))
