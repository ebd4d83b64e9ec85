"""DeterministicBandit lanes on the device and the reference's universal agent test on them:
`train_deterministic_bandit` (src/agents/testing.rs:14-64) — bandit [0, 1], 10 update periods, then the evaluation actor
must pick arm 1 in at least 90 % of 1,000 steps — over the reference's matrix of policy rules, critics and modules
(src/torch/agents/actor_critic.rs:292-332: Adam 0.1 and one optimisation step for PPO / REINFORCE / the critic, TRPO's
defaults).  The reference gives each period >= 25 steps (`HistoryDataBound::new(25, 1)`); here a period is one step of
32 lanes."""
import numpy as np
import pytest

import oracle as O
import relearn_amd as ra

pytestmark = pytest.mark.gpu


def test_bandit_lanes_bit_exact(engine):
    n, T = 96, 7
    env = ra.BanditEnv(engine, n, values=(0.25, 1.5), seed_env=5, seed_actor=6)
    sim = O.BanditLaneSim(n, values=(0.25, 1.5), seed_env=5, seed_actor=6)
    assert (env.D, env.A) == (5, 2)
    assert np.array_equal(env.observe(), sim.observe())
    rng = np.random.default_rng(0)
    for _ in range(3):
        a = rng.integers(0, 2, n).astype(np.uint8)
        got, want = env.step(a), sim.step(a)
        for g, w in zip(got[:3], want[:3]):
            assert np.array_equal(g, w)
    pol = ra.Mlp(engine, 5, 128, 2)
    pol.init(3)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    want = sim.rollout_mlp(O.MlpShape(5, 128, 2), pol.get_params(), T)
    got = traj.read_all()
    for k in ("obs", "action", "reward", "flag"):
        assert np.array_equal(got[k], want[k]), k
    assert (got["flag"] == O.TERMINATE).all()
    assert np.array_equal(got["reward"], np.where(got["action"] == 1, np.float32(1.5), np.float32(0.25)))


def adam(module, lr):
    cfg = ra.adam_config_default()
    cfg.learning_rate = lr
    return ra.Adam(module, cfg)


def train_deterministic_bandit(engine, module, policy_rule, critic, num_periods=10, threshold=0.9):
    n = 32
    env = ra.BanditEnv(engine, n, values=(0.0, 1.0), seed_env=18, seed_actor=19)  # DeterministicBandit::from_values([0, 1])
    make = (lambda out: ra.Mlp(engine, 5, 128, out)) if module == "mlp" else (lambda out: ra.GruMlp(engine, 5, out))
    pol = make(2)
    pol.init(19)
    popt = adam(pol, 0.1) if policy_rule != "trpo" else None
    ppo = ra.ppo_config_default()
    ppo.opt_steps_per_update = 1
    cri, copt, vcfg = None, None, None
    if critic != "r2g":
        cri = make(1)
        cri.init(20)
        copt = adam(cri, 0.1)
        vcfg = ra.values_opt_config_default()
        vcfg.opt_steps_per_update = 1
        vcfg.target = ra.VALUE_TARGET_ONE_STEP_TD if critic == "td" else ra.VALUE_TARGET_REWARD_TO_GO
        vcfg.discount_factor = 0.99  # min(max_discount_factor, the env's 1.0) (critics/opt.rs:73)
    traj = ra.Trajectory(engine, n, 1, 5)
    for _ in range(num_periods):
        ra.rollout(env, pol, traj)
        if cri is None:
            ra.reward_to_go(traj, 1.0)  # RewardToGo with the env's own discount factor (bandits.rs:52-54)
        else:
            ra.gae(traj, cri, 0.99, 0.95)
        if policy_rule == "trpo":
            ra.trpo_update(pol, traj)
        elif policy_rule == "ppo":
            ra.ppo_update(pol, popt, traj, ppo)
        else:
            ra.reinforce_update(pol, popt, traj)
        if cri is not None:
            ra.values_opt_update(cri, copt, traj, vcfg)
    # eval_deterministic_bandit: 1,000 steps of the (sampling) evaluation actor
    ev = ra.Trajectory(engine, n, 32, 5)
    ra.rollout(env, pol, ev)
    actions = ev.read(ra.TRAJ_ACTION).reshape(-1)[:1000]
    assert (actions == 1).sum() >= int(1000 * threshold), ((actions == 1).sum(), module, policy_rule, critic)


@pytest.mark.parametrize("module", ["mlp", "gru"])
@pytest.mark.parametrize("policy_rule", ["reinforce", "ppo", "trpo"])
def test_learns_deterministic_bandit_r2g(engine, module, policy_rule):
    train_deterministic_bandit(engine, module, policy_rule, "r2g")


@pytest.mark.parametrize("module", ["mlp", "gru"])
@pytest.mark.parametrize("policy_rule", ["reinforce", "ppo", "trpo"])
@pytest.mark.parametrize("target", ["rtg", "td"])
def test_learns_deterministic_bandit_values_gae(engine, module, policy_rule, target):
    train_deterministic_bandit(engine, module, policy_rule, target)
