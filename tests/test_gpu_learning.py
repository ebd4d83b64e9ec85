"""End-to-end behaviour on the GPU: the agents the hot path trains actually learn CartPole.  The reference's own
integration tests train every agent on a deterministic bandit and ask for >= 90 % optimal pulls
(src/agents/testing.rs:14-64), and never run TRPO or PPO end to end (its `trpo()` / `ppo()` test helpers
return the REINFORCE config); these tests close that gap on the workload of the headline metric.  Episode length is the
return of CartPole (reward 1 per step, src/envs/cartpole.rs:140); a random policy lasts ~22 steps, the visible step limit
is 500."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")


def mean_episode_length(traj):
    """steps per finished episode within the period (lanes persist across periods, so this is a moving estimate)"""
    flag = traj.read(ra.TRAJ_FLAG)
    return flag.size / max(int((flag != 0).sum()), 1)


def actor_critic(engine, policy_update, periods, n=4096, T=128, critic_steps=20):
    env = ra.CartPoleEnv(engine, n, max_steps=500, seed_env=10, seed_actor=11)
    pol, cri = ra.Mlp(engine, 5, 128, 2), ra.Mlp(engine, 5, 128, 1)
    pol.init(12)
    cri.init(13)
    copt = ra.Adam(cri)
    traj = ra.Trajectory(engine, n, T, 5)
    lengths = []
    for _ in range(periods):
        ra.rollout(env, pol, traj)
        lengths.append(mean_episode_length(traj))
        ra.gae(traj, cri, 0.99, 0.95)
        policy_update(pol, traj)
        ra.critic_update(cri, copt, traj, critic_steps)
    ra.rollout(env, pol, traj)
    lengths.append(mean_episode_length(traj))
    return lengths


def test_trpo_learns_cartpole(engine):
    stats = []

    def update(pol, traj):
        st = ra.trpo_update(pol, traj)
        stats.append(st)

    lengths = actor_critic(engine, update, 25)
    assert lengths[0] < 30, lengths
    assert lengths[-1] > 4 * lengths[0], lengths
    ok = [s for s in stats if s.status == ra.OPT_OK]
    assert len(ok) >= 20  # the line search accepts a step in (nearly) every period ...
    assert all(s.constraint_val_final <= 0.01 and s.loss_final < s.loss_initial for s in ok)  # ... inside the region
    assert stats[-1].entropy < stats[0].entropy  # the policy commits


def test_ppo_learns_cartpole(engine):
    opt = {}

    def update(pol, traj):
        if "adam" not in opt:
            acfg = ra.adam_config_default()
            acfg.learning_rate = 3e-3
            opt["adam"] = ra.Adam(pol, acfg)
        ra.ppo_update(pol, opt["adam"], traj, ra.ppo_config_default())

    lengths = actor_critic(engine, update, 25)
    assert lengths[0] < 30 and lengths[-1] > 3 * lengths[0], lengths


def test_dqn_learns_cartpole(engine):
    n = 1024
    env = ra.CartPoleEnv(engine, n, max_steps=500, seed_env=20, seed_actor=21)
    q = ra.Mlp(engine, 5, 128, 2)
    q.init(22)
    cfg = ra.dqn_config_default()
    cfg.minibatch_steps, cfg.opt_steps_per_update, cfg.buffer_capacity = 20000, 20, 4096
    cfg.update_first, cfg.update_rest, cfg.exploration_period = n * 64, n * 32, n * 32 * 20
    cfg.discount_factor = 0.99
    for i in range(8):
        cfg.agent_key[i] = 100 + i
    dqn = ra.Dqn(env, q, ra.Adam(q), cfg)
    first = None
    for _ in range(40):
        m, _ = dqn.min_update_size()
        st = dqn.collect((m + n - 1) // n)
        if first is None:  # 64 steps per lane with exploration rate 1.0: the random policy
            first = st.steps / st.episodes_ended
        dqn.update()
    # every collection ends with an Interrupt in every lane (the horizon rule), so the short training collections cap the
    # episode length they can show at their own length; measure with one long collection (exploration rate 0.1)
    assert abs(dqn.exploration_rate(True) - 0.1) < 1e-6
    st = dqn.collect(1000)
    final = st.steps / st.episodes_ended
    assert first < 30 and final > 4 * first, (first, final)
