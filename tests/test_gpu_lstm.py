"""GPU parity tests of the LSTM chain (`ChainConfig<LstmConfig, MlpConfig>`: Lstm = RnnBase<LstmImpl>,
src/torch/modules/seq/rnn/lstm.rs:12-51) through the C ABI against oracle/seq_impl.inc (cell =
LSTM) on the same seeds.  Bars as for the GRU chain (tests/test_gpu_gru.py): bit-exact for the initialisation,
rollouts on every env kind, teacher-forced outputs, successor outputs, values / advantages / returns; fp32 tolerances,
stated below, for gradients."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

PS, CS = O.LstmShape(5, 128, 128, 2), O.LstmShape(5, 128, 128, 1)
L = O.lib()


def test_lstm_init_bit_exact(engine):
    for shape, seed in ((PS, 5), (CS, 6)):
        m = ra.LstmMlp(engine, 5, shape.out_dim)
        m.init(seed)
        assert m.P == L.oracle_gru_num_params(shape) == 4 * 128 * 5 + 4 * 128 * 128 + 8 * 128 + 128 * 128 + 128 + \
            shape.out_dim * 128 + shape.out_dim
        assert np.array_equal(m.get_params(), O.gru_init(shape, seed))
    with pytest.raises(ra.RelearnError) as e:
        ra.LstmMlp(engine, 5, 2, lstm_hidden=64)
    assert e.value.code == ra.ERR_BUILD_AGENT


@pytest.mark.parametrize("kind", ["memory", "chain", "cartpole"])
def test_rollout_bit_exact(engine, kind):
    n, T = 96, 41
    if kind == "memory":
        env, sim = ra.MemoryEnv(engine, n, seed_env=3, seed_actor=4), O.MemoryLaneSim(n, seed_env=3, seed_actor=4)
    elif kind == "chain":
        env = ra.ChainEnv(engine, n, max_steps=9, seed_env=3, seed_actor=4)
        sim = O.ChainLaneSim(n, max_steps=9, seed_env=3, seed_actor=4)
    else:
        env = ra.CartPoleEnv(engine, n, max_steps=25, seed_env=5, seed_actor=6)
        sim = O.LaneSim(n, max_steps=25, seed_env=5, seed_actor=6)
    pol = ra.LstmMlp(engine, 5, 2)
    pol.init(11)
    traj = ra.Trajectory(engine, n, T, 5)
    for period in range(2):  # lanes, step counters and stream positions persist across periods
        ra.rollout(env, pol, traj)
        want = sim.rollout_gru(PS, pol.get_params(), T)
        got = traj.read_all()
        for k in ("obs", "action", "reward", "flag"):
            assert np.array_equal(got[k], want[k]), (kind, period, k)
        m = want["flag"] == O.INTERRUPT
        assert np.array_equal(got["term_obs"][:, m], want["term_obs"][:, m])
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)
    assert 0.2 < want["action"].mean() < 0.8 and (want["flag"] != O.CONTINUE).any()


@pytest.mark.parametrize("max_steps", [100, 9])
def test_seq_forward_and_gae_bit_exact(engine, max_steps):
    n, T = 64, 30
    env = ra.ChainEnv(engine, n, max_steps=max_steps, seed_env=3, seed_actor=4)
    sim = O.ChainLaneSim(n, max_steps=max_steps, seed_env=3, seed_actor=4)
    pol, cri = ra.LstmMlp(engine, 5, 2), ra.LstmMlp(engine, 5, 1)
    pol.init(21)
    cri.init(22)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    want = sim.rollout_gru(PS, pol.get_params(), T)
    for mod, shape in ((pol, PS), (cri, CS)):
        out_d, succ_d = mod.seq_forward(traj)
        out_o, succ_o = O.gru_seq_forward(shape, mod.get_params(), want)
        assert np.array_equal(out_d, out_o) and np.array_equal(succ_d, succ_o)
    ra.gae(traj, cri, 0.95, 0.9)
    v, s = O.gru_seq_forward(CS, cri.get_params(), want)
    adv_o, rtg_o = O.seq_gae(v[0], s[0], want, np.float32(0.95), np.float32(0.9))
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), adv_o) and np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)
    if max_steps < T:
        assert np.count_nonzero(succ_o) > 0  # cut episodes evaluate their successor observation from (h, c)
