"""GPU parity of the first-order policy updates (PPO, REINFORCE) and the critic-free RewardToGo advantage —
through the C ABI against the oracle restatement (itself pinned by torch autograd, tests/test_oracle_ppo.py).
Tolerances: sums over samples in a different order (as tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

H = 128
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)
L = O.lib()


@pytest.fixture(params=[0, 1], ids=["kernels-best", "kernels-v1"])
def variant(engine, request):
    engine.set_kernel_variant(request.param)
    yield request.param
    engine.set_kernel_variant(0)


def setup(engine, n=512, T=64, max_steps=30, lr=1e-3):
    env = ra.CartPoleEnv(engine, n, max_steps=max_steps)
    sim = O.LaneSim(n, max_steps=max_steps)
    policy = ra.Mlp(engine, 5, H, 2)
    policy.init(2)
    critic = ra.Mlp(engine, 5, H, 1)
    critic.init(3)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, policy, traj)
    want = sim.rollout(PS, policy.get_params(), T)
    ra.gae(traj, critic, 0.99, 0.95)
    x, a = O.flat_samples(want)
    adv = np.ascontiguousarray(traj.read(ra.TRAJ_ADVANTAGES).reshape(-1))
    acfg = ra.adam_config_default()
    acfg.learning_rate = lr
    opt = ra.Adam(policy, acfg)
    ocfg = O.AdamCfg()
    L.oracle_adam_cfg_default(C.byref(ocfg))
    ocfg.lr = lr
    return policy, opt, traj, want, x, a, adv, ocfg


@pytest.mark.parametrize("lr,steps", [(1e-3, 10), (2e-2, 8)], ids=["default", "clipping-active"])
def test_ppo_update(engine, variant, lr, steps):
    policy, opt, traj, want, x, a, adv, ocfg = setup(engine, lr=lr)
    p = policy.get_params().copy()
    cfg = ra.ppo_config_default()
    assert (cfg.opt_steps_per_update, cfg.clip_distance) == (10, 0.2)
    cfg.opt_steps_per_update = steps
    st, losses_d = ra.ppo_update(policy, opt, traj, cfg, want_losses=True)
    ost = L.oracle_adam_new(len(p))
    losses_o = np.zeros(steps, np.float32)
    ent = C.c_float()
    L.oracle_ppo_update_f32(PS, O.f32p(p), ost, C.byref(ocfg), O.f32p(x), O.i64p(a), O.f32p(adv), len(a), steps,
                            0.2, O.f32p(losses_o), C.byref(ent))
    L.oracle_adam_free(ost)
    assert abs(st.entropy - ent.value) < 1e-5
    assert st.steps == steps and st.loss_first == losses_d[0] and st.loss_last == losses_d[-1]
    scale = max(1.0, np.abs(losses_o).max())
    assert np.max(np.abs(losses_d - losses_o)) < 2e-5 * scale, (losses_d, losses_o)
    # Adam normalises by sqrt(v): where |g| is at rounding level a parameter can move by lr either way
    assert np.abs(policy.get_params() - p).max() < (2e-5 if lr < 1e-2 else 5e-3)
    assert np.mean(np.abs(policy.get_params() - p) < 2e-5) > 0.97
    if lr > 1e-2:
        # the clip was active: ratios of the final policy leave [0.8, 1.2] for a visible share of samples
        lp0 = np.zeros(len(a), np.float32)
        lp1 = np.zeros(len(a), np.float32)
        L.oracle_policy_logp_f32(PS, O.f32p(O.mlp_init(PS, 2)), O.f32p(x), O.i64p(a), len(a), O.f32p(lp0), None)
        L.oracle_policy_logp_f32(PS, O.f32p(policy.get_params()), O.f32p(x), O.i64p(a), len(a), O.f32p(lp1), None)
        r = np.exp(lp1 - lp0)
        assert ((r < 0.8) | (r > 1.2)).mean() > 0.02
    assert losses_d[-1] < losses_d[0]


def test_reinforce_update(engine, variant):
    policy, opt, traj, want, x, a, adv, ocfg = setup(engine)
    p = policy.get_params().copy()
    st = ra.reinforce_update(policy, opt, traj)
    ost = L.oracle_adam_new(len(p))
    loss, ent = C.c_float(), C.c_float()
    L.oracle_reinforce_update_f32(PS, O.f32p(p), ost, C.byref(ocfg), O.f32p(x), O.i64p(a), O.f32p(adv), len(a),
                                  C.byref(loss), C.byref(ent))
    L.oracle_adam_free(ost)
    assert abs(st.loss_first - loss.value) <= 1e-5 * max(1.0, abs(loss.value))
    assert abs(st.entropy - ent.value) < 1e-5
    # first Adam step: every parameter moves by lr * sign(g) (up to eps): identical unless g is at rounding level
    assert np.mean(np.abs(policy.get_params() - p) < 1e-6) > 0.99


def test_reward_to_go_critic_bit_exact(engine):
    policy, opt, traj, want, x, a, adv, ocfg = setup(engine, n=384, T=96, max_steps=25)
    ra.reward_to_go(traj, 0.99)
    _, _, rtg_o = O.lanes_gae(CS, O.mlp_init(CS, 3), want, np.float32(0.99), np.float32(0.95))
    assert np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), rtg_o)
