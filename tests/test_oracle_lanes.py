"""The lane-model restatement (what the HIP engine implements) against the reference-structured restatement
(VecBuffer -> LazyHistoryFeatures -> packed GAE), plus environment invariants and the chain/tabular-Q
plumbing configuration.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

L = O.lib()
H = 32
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)


def _rollout(n=96, T=40, max_steps=13, seed=2):
    sim = O.LaneSim(n, max_steps=max_steps, seed_env=4, seed_actor=9)
    return sim, sim.rollout(PS, O.mlp_init(PS, seed), T, threads=4)


def test_lane_gae_equals_reference_packed_gae():
    """Lane-major GAE/RTG (engine layout; horizon cut kept as Interrupt(obs[T])) must equal the reference
    pipeline: episodes -> VecBuffer -> sort by length -> packed extended observations -> masked values ->
    trim_end/trim_start -> discounted_cumsum_from_end — bit for bit."""
    sim, traj = _rollout()
    D, T1, n = traj["obs"].shape
    T = T1 - 1
    cp = O.mlp_init(CS, 3)
    gamma, lam = np.float32(0.99), np.float32(0.95)
    v, adv, rtg = O.lanes_gae(CS, cp, traj, gamma, lam)
    idx = np.zeros(n * T, np.uint64)
    buf = L.oracle_lanes_to_vecbuffer(n, T, D, O.f32p(traj["obs"]), O.u8p(traj["action"]), O.f32p(traj["reward"]),
                                      O.u8p(traj["flag"]), O.f32p(traj["term_obs"]), 1, O.u64p(idx))
    assert buf.contents.len == n * T
    arr = (C.POINTER(O.VecBuffer) * 1)(buf)
    feat = L.oracle_features_from_buffers(arr, 1)
    ns = feat.contents.n_steps
    adv_p, rtg_p = np.zeros(ns, np.float32), np.zeros(ns, np.float32)
    L.oracle_gae_packed(CS, O.f32p(cp), feat, gamma, lam, O.f32p(adv_p), None)
    L.oracle_reward_to_go_packed(feat, gamma, O.f32p(rtg_p))
    src = np.array([feat.contents.src_index[i] for i in range(ns)], np.int64)
    lane_t = idx[src].astype(np.int64)  # packed position -> lane * T + t
    lane, t = lane_t // T, lane_t % T
    assert np.array_equal(adv_p, adv[t, lane])
    assert np.array_equal(rtg_p, rtg[t, lane])
    # StepValueTarget::OneStepTd (critics/mod.rs:139-150): the lane form against the packed reference pipeline
    td_p = np.zeros(ns, np.float32)
    L.oracle_one_step_values_packed(CS, O.f32p(cp), feat, gamma, O.f32p(td_p))
    td = O.lanes_one_step_targets(CS, cp, traj, gamma)
    assert np.array_equal(td_p, td[t, lane])
    assert (traj["flag"] == O.TERMINATE).any() and (traj["flag"] == O.INTERRUPT).any()  # all three successor kinds
    # delta_t = target_t - V_t: the TD target is the residual of temporal_differences plus the value (same roundings)
    assert np.array_equal((td - v[:-1]).astype(np.float32)[traj["flag"] != O.CONTINUE],
                          adv[traj["flag"] != O.CONTINUE])
    # the flat sample arrays the update kernels use hold the same multiset of samples as the packed features
    x, a = O.flat_samples(traj)
    packed_obs = np.array([[feat.contents.obs[i * D + d] for d in range(D)] for i in range(ns)], np.float32)
    assert np.array_equal(packed_obs, x[t * n + lane])
    L.oracle_features_free(feat)
    L.oracle_vecbuffer_free(buf)


def test_reference_end_of_buffer_rule_drops_the_dangling_step():
    """With keep_last = 0 each lane is finalised like a reference experience thread (buffers/mod.rs:237-261):
    a trailing Continue step is dropped and its observation becomes the previous step's Interrupt successor."""
    sim, traj = _rollout(n=16, T=20, max_steps=500)
    D, T1, n = traj["obs"].shape
    T = T1 - 1
    buf = L.oracle_lanes_to_vecbuffer(n, T, D, O.f32p(traj["obs"]), O.u8p(traj["action"]), O.f32p(traj["reward"]),
                                      O.u8p(traj["flag"]), O.f32p(traj["term_obs"]), 0, None)
    dangling = int((traj["flag"][T - 1] == O.CONTINUE).sum())
    assert buf.contents.len == n * T - dangling
    nxt = np.zeros(buf.contents.len, np.uint8)
    nobs = np.zeros((buf.contents.len, D), np.float32)
    L.oracle_vecbuffer_steps(buf, None, None, None, O.u8p(nxt), O.f32p(nobs))
    assert nxt[-1] != O.CONTINUE
    L.oracle_vecbuffer_free(buf)


def test_cartpole_invariants_and_libm_agreement():
    """check_structured_env-style invariants (envs/testing.rs:23-57) + the engine's sincos contract agrees with
    platform libm to the last ulps over whole episodes' worth of single steps."""
    n = 512
    sim = O.LaneSim(n, max_steps=500, seed_env=11)
    sim_libm = O.LaneSim(n, max_steps=500, seed_env=11, env=O.cartpole_default(use_libm=True))
    rng = np.random.default_rng(0)
    for _ in range(300):
        st, nv, rem, rc = sim.get_state()
        sim_libm.set_state(st, nv, rem, rc)
        a = rng.integers(0, 2, n).astype(np.uint8)
        r, f, obs, term = sim.step(a)
        r2, f2, obs2, term2 = sim_libm.step(a)
        assert np.all(r == 1.0)
        assert np.all(np.abs(obs[0]) <= 2.4) and np.all(np.abs(obs[2]) <= 12 * np.pi / 180)
        assert np.all((obs[4] >= 0) & (obs[4] <= 1))
        same = (f == f2)
        assert same.mean() > 0.999
        cont = same & (f == O.CONTINUE)
        st_a, st_b = sim.get_state()[0], sim_libm.get_state()[0]
        assert np.allclose(st_a[:, cont], st_b[:, cont], rtol=0, atol=1e-15)
    st, nv, rem, rc = sim.get_state()
    assert rc.max() > 1  # episodes ended and restarted


def test_rollout_statistics_and_determinism():
    sim, traj = _rollout(n=64, T=64, max_steps=500)
    sim2, traj2 = _rollout(n=64, T=64, max_steps=500)
    for k in traj:
        assert np.array_equal(traj[k], traj2[k])
    # a freshly initialised policy is close to uniform
    assert 0.3 < traj["action"].mean() < 0.7
    assert (traj["flag"] == O.TERMINATE).sum() > 0
    # sharding invariance of the lane streams
    simh = O.LaneSim(32, max_steps=500, seed_env=4, seed_actor=9, lane_offset=32)
    th = simh.rollout(PS, O.mlp_init(PS, 2), 64, threads=2)
    assert np.array_equal(th["action"], traj["action"][:, 32:])
    assert np.array_equal(th["obs"], traj["obs"][:, :, 32:])


def test_chain_tabular_q_plumbing_config():
    """configs[0]: examples/chain-tabular-q.rs through train_parallel — deterministic given the thread count,
    learns the optimal policy of the 5-state chain (always Right: Q[s,1] > Q[s,0])."""
    q = np.zeros((5, 2), np.float64)
    cnt = np.zeros((5, 2), np.uint64)
    tot = C.c_uint64()
    L.oracle_chain_tabular_q_train(0, 4, 10, 10000, O.f64p(q), O.u64p(cnt), C.byref(tot))
    q2, cnt2, tot2 = np.zeros_like(q), np.zeros_like(cnt), C.c_uint64()
    L.oracle_chain_tabular_q_train(0, 4, 10, 10000, O.f64p(q2), O.u64p(cnt2), C.byref(tot2))
    assert np.array_equal(q, q2) and np.array_equal(cnt, cnt2)
    assert tot.value == 4 * 10 * (10000 - 1) == int(cnt.sum())
    assert np.all(q[:, 1] > q[:, 0])
    acts = np.zeros(10000, np.int32)
    total = L.oracle_chain_tabular_q_eval(O.f64p(q), 0, 10000, O.i32p(acts))
    assert np.all(acts == 1)
    assert total / 10000 > 3.0  # optimal policy earns ~3.6 per step with slip 0.2; greedy-left earns 2.0
    # step_update restated (tabular.rs:159-180): 1/n step size
    t = L.oracle_tabular_q_new(2, 2, 0.5, 0.0)
    vals, cnts = np.zeros((2, 2), np.float64), np.zeros((2, 2), np.uint64)
    L.oracle_tabular_q_step_update(t, 0, 1, 4.0, O.TERMINATE, 0)
    L.oracle_tabular_q_read(t, O.f64p(vals), O.u64p(cnts))
    assert vals[0, 1] == 4.0 and cnts[0, 1] == 1  # first visit: weight 1
    L.oracle_tabular_q_step_update(t, 0, 1, 2.0, O.TERMINATE, 0)
    L.oracle_tabular_q_read(t, O.f64p(vals), O.u64p(cnts))
    assert vals[0, 1] == 3.0 and cnts[0, 1] == 2  # second visit: the running mean of 4 and 2
    assert np.count_nonzero(vals) == 1
    L.oracle_tabular_q_free(t)


def test_chain_env_transitions():
    env = O.Chain()
    L.oracle_chain_default(C.byref(env))
    r = O.Prng()
    L.oracle_prng_seed_from_u64(C.byref(r), 3)
    state = C.c_uint64(0)
    rew = C.c_double()
    slips = 0
    for i in range(4000):
        s0 = state.value
        a = i % 2
        assert L.oracle_chain_step(C.byref(env), C.byref(state), a, C.byref(r), C.byref(rew)) == O.CONTINUE
        moved_left = state.value == 0 and rew.value == 2.0
        if moved_left != (a == 0):
            slips += 1
        if not moved_left:
            assert (state.value == min(s0 + 1, 4)) and rew.value == (10.0 if s0 == 4 else 0.0)
    assert abs(slips / 4000 - 0.2) < 0.03
