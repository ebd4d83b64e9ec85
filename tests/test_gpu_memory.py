"""GPU parity tests of MemoryGame lanes (src/envs/memory.rs:24-115; SURVEY §8f rank 4: "a sharper partially-observed
benchmark" for the recurrent policy) through the C ABI against the oracle on the same seeds — bit-exact states,
stream positions, observations, rewards, flags and rollouts — plus the behavioural check the reference applies to its
agents (agents/testing.rs:14-64: the trained agent picks the right arm >= 90 % of the time): only a policy with
memory can beat 50 % here, the GRU policy does, the feed-forward policy cannot."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

PS = O.GruShape(5, 128, 128, 2)


def pair(engine, n, **kw):
    return ra.MemoryEnv(engine, n, **kw), O.MemoryLaneSim(n, **kw)


def states_equal(env, sim):
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("limit,max_steps", [(ra.LIMIT_NONE, 0), (ra.LIMIT_LATENT, 3), (ra.LIMIT_LATENT, 4)])
def test_memory_env_bit_exact(engine, limit, max_steps):
    n = 3000
    env, sim = pair(engine, n, max_steps=max_steps, limit=limit, lane_offset=123, seed_env=9)
    assert (env.D, env.A) == (5, 2)
    states_equal(env, sim)  # initial states and the words their rejection loops consumed
    assert np.array_equal(env.observe(), sim.observe())
    rng = np.random.default_rng(2)
    seen = set()
    for t in range(30):
        a = rng.integers(0, 2, n).astype(np.uint8)
        rd, fd, od, td = env.step(a)
        ro, fo, oo, to = sim.step(a)
        assert np.array_equal(rd, ro) and np.array_equal(fd, fo) and np.array_equal(od, oo)
        m = fo == O.INTERRUPT
        assert np.array_equal(td[:, m], to[:, m])
        seen |= set(np.unique(fo).tolist())
        states_equal(env, sim)
    assert set(np.unique(ro)) <= {-1.0, 0.0, 1.0}
    if max_steps == 3:
        assert seen == {O.CONTINUE, O.INTERRUPT}  # cut before the answer step, every time
    else:
        assert seen == {O.CONTINUE, O.TERMINATE}
    env.reset()  # Environment::initial_state again: continues each lane's stream
    sim.reset()
    states_equal(env, sim)


def test_unsupported_sizes_are_refused(engine):
    with pytest.raises(ra.RelearnError) as e:
        ra.MemoryEnv(engine, 64, num_actions=2, history_len=1)  # MemoryGame::default has 3 observation features
    assert e.value.code == ra.ERR_BUILD_ENV


@pytest.mark.parametrize("n,T", [(64, 37), (96, 16)])
def test_rollouts_bit_exact(engine, n, T):
    env, sim = pair(engine, n, seed_env=3, seed_actor=4)
    pol = ra.GruMlp(engine, 5, 2)
    pol.init(11)
    traj = ra.Trajectory(engine, n, T, 5)
    for period in range(2):
        ra.rollout(env, pol, traj)
        want = sim.rollout_gru(PS, pol.get_params(), T)
        got = traj.read_all()
        for k in ("obs", "action", "reward", "flag"):
            assert np.array_equal(got[k], want[k]), (period, k)
        states_equal(env, sim)
    assert (want["flag"] == O.TERMINATE).sum() >= n * (T // 4 - 1)
    # feed-forward policy on the same lanes (continuing streams)
    ms = O.MlpShape(5, 128, 2)
    mlp = ra.Mlp(engine, 5, 128, 2)
    mlp.init(12)
    ra.rollout(env, mlp, traj)
    want = sim.rollout_mlp(ms, mlp.get_params(), T)
    got = traj.read_all()
    for k in ("obs", "action", "reward", "flag"):
        assert np.array_equal(got[k], want[k]), k
    states_equal(env, sim)


def answer_accuracy(traj):
    r = traj.read(ra.TRAJ_REWARD)
    return float((r == 1.0).sum()) / float((r != 0.0).sum())


def train(engine, pol, periods, n=2048, T=32):
    env = ra.MemoryEnv(engine, n, seed_env=21, seed_actor=22)
    traj = ra.Trajectory(engine, n, T, 5)
    acfg = ra.adam_config_default()
    acfg.learning_rate = 3e-3
    opt = ra.Adam(pol, acfg)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 5
    acc = []
    for _ in range(periods):
        ra.rollout(env, pol, traj)
        acc.append(answer_accuracy(traj))
        ra.reward_to_go(traj, 1.0)  # MemoryGame::discount_factor is 1.0 (memory.rs:74-76); critic-free advantage
        ra.ppo_update(pol, opt, traj, cfg)
    ra.rollout(env, pol, traj)
    acc.append(answer_accuracy(traj))
    return acc


def test_only_the_recurrent_policy_learns_the_memory_game(engine):
    gru = ra.GruMlp(engine, 5, 2)
    gru.init(5)
    acc = train(engine, gru, 25)
    assert 0.4 < acc[0] < 0.6 and acc[-1] >= 0.9, acc
    # the answer step's observation (state 4) carries no information about the initial state: a feed-forward policy
    # stays at chance whatever it learns
    mlp = ra.Mlp(engine, 5, 128, 2)
    mlp.init(6)
    acc = train(engine, mlp, 10)
    assert all(0.45 < a < 0.55 for a in acc), acc
