"""GPU parity tests of the recurrent configuration (BASELINE.json configs[4]): Chain lanes under a latent step
limit and the GRU -> ReLU -> MLP module, through the C ABI against oracle/seq.c on the same seeds.

Bars: bit-exact for the env, the initialisation, rollouts (observations, actions, rewards, flags), teacher-forced
module outputs and successor outputs, values / advantages / returns (the MFMA accumulation is an exact sequential
fma chain, restated as such by the oracle); fp32 tolerances, stated below, for gradients (sums over samples in a
different order)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

PS, CS = O.GruShape(5, 128, 128, 2), O.GruShape(5, 128, 128, 1)
L = O.lib()


def chain_pair(engine, n, max_steps=100, limit=ra.LIMIT_LATENT, lane_offset=0, seed_env=3, seed_actor=4):
    env = ra.ChainEnv(engine, n, max_steps=max_steps, limit=limit, lane_offset=lane_offset, seed_env=seed_env,
                      seed_actor=seed_actor)
    sim = O.ChainLaneSim(n, max_steps=max_steps, limit=limit, lane_offset=lane_offset, seed_env=seed_env,
                         seed_actor=seed_actor)
    return env, sim


def test_chain_env_bit_exact(engine):
    env, sim = chain_pair(engine, 1000, max_steps=7, lane_offset=77)
    assert (env.D, env.A) == (5, 2)
    rng = np.random.default_rng(1)
    assert np.array_equal(env.observe(), sim.observe())
    for t in range(40):
        a = rng.integers(0, 2, 1000).astype(np.uint8)
        rd, fd, od, td = env.step(a)
        ro, fo, oo, to = sim.step(a)
        assert np.array_equal(rd, ro) and np.array_equal(fd, fo) and np.array_equal(od, oo)
        m = fo == O.INTERRUPT
        assert np.array_equal(td[:, m], to[:, m])
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)
    assert (fo == O.INTERRUPT).sum() == 0 and set(np.unique(ro)) <= {0.0, 2.0, 10.0}


def test_gru_init_bit_exact(engine):
    for shape, seed in ((PS, 5), (CS, 6)):
        m = ra.GruMlp(engine, 5, shape.out_dim)
        m.init(seed)
        assert m.P == L.oracle_gru_num_params(shape)
        assert np.array_equal(m.get_params(), O.gru_init(shape, seed))


@pytest.mark.parametrize("n,T,max_steps", [(64, 40, 100), (96, 50, 9)])
def test_rollout_bit_exact(engine, n, T, max_steps):
    env, sim = chain_pair(engine, n, max_steps=max_steps)
    pol = ra.GruMlp(engine, 5, 2)
    pol.init(11)
    traj = ra.Trajectory(engine, n, T, 5)
    for period in range(2):  # lanes, step counters and stream positions persist across periods
        ra.rollout(env, pol, traj)
        want = sim.rollout_gru(PS, pol.get_params(), T)
        got = traj.read_all()
        for k in ("obs", "action", "reward", "flag"):
            assert np.array_equal(got[k], want[k]), (period, k)
        m = want["flag"] == O.INTERRUPT
        assert np.array_equal(got["term_obs"][:, m], want["term_obs"][:, m])
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)
    assert 0.2 < want["action"].mean() < 0.8  # both actions occur
    if max_steps < T:
        assert m.any()


@pytest.mark.parametrize("max_steps", [100, 9])
def test_seq_forward_and_gae_bit_exact(engine, max_steps):
    n, T = 96, 45
    env, sim = chain_pair(engine, n, max_steps=max_steps)
    pol, cri = ra.GruMlp(engine, 5, 2), ra.GruMlp(engine, 5, 1)
    pol.init(11)
    cri.init(12)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    want = sim.rollout_gru(PS, pol.get_params(), T)
    lo_d, _ = pol.seq_forward(traj, want_succ=False)
    lo_o, _ = O.gru_seq_forward(PS, pol.get_params(), want, want_succ=False)
    assert np.array_equal(lo_d, lo_o)
    v_d, s_d = cri.seq_forward(traj)
    v_o, s_o = O.gru_seq_forward(CS, cri.get_params(), want)
    assert np.array_equal(v_d, v_o) and np.array_equal(s_d, s_o)
    assert np.count_nonzero(s_o[0, T - 1]) == n  # horizon cut: every lane bootstraps from its successor
    ra.gae(traj, cri, 0.95, 0.9)
    adv_o, rtg_o = O.seq_gae(v_o[0], s_o[0], want, np.float32(0.95), np.float32(0.9))
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), adv_o)
    assert np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)
