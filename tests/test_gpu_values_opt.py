"""ValuesOpt::update with its StepValueTarget (src/torch/agents/critics/opt.rs:100-126; critics/mod.rs:139-150,
203-229) through the C ABI: the targets are bit-exact against the oracle (integer-free but order-fixed f32 arithmetic:
r + gamma * V_next with Terminate -> 0, Interrupt -> V(successor)), the optimisation steps follow the oracle's within
the tolerances of the existing critic tests.  Feed-forward and recurrent (GRU, LSTM) critics."""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

L = O.lib()
H = 128
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)


def _cartpole(engine, n, T, max_steps, seed=2):
    env = ra.CartPoleEnv(engine, n, max_steps=max_steps)
    sim = O.LaneSim(n, max_steps=max_steps)
    policy = ra.Mlp(engine, 5, H, 2)
    policy.init(seed)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, policy, traj)
    return traj, sim.rollout(PS, policy.get_params(), T)


@pytest.mark.parametrize("variant", [0, 1], ids=["kernels-best", "kernels-v1"])
def test_one_step_td_targets_and_update_mlp(engine, variant):
    engine.set_kernel_variant(variant)
    try:
        n, T, steps = 384, 64, 12
        traj, want = _cartpole(engine, n, T, max_steps=25)
        assert (want["flag"] == O.INTERRUPT).any() and (want["flag"] == O.TERMINATE).any()
        critic = ra.Mlp(engine, 5, H, 1)
        critic.init(3)
        cp = critic.get_params()
        cfg = ra.values_opt_config_default()
        cfg.opt_steps_per_update, cfg.target, cfg.discount_factor = steps, ra.VALUE_TARGET_ONE_STEP_TD, 0.97
        opt = ra.Adam(critic)
        ra.gae(traj, critic, 0.97, 0.95)  # RL_TRAJ_RETURNS of the initial critic (used at the end of this test)
        st, losses_d = ra.values_opt_update(critic, opt, traj, cfg, want_losses=True)
        # targets: computed once from the critic as it stood BEFORE the first step (tch::no_grad, opt.rs:101-104)
        td_o = O.lanes_one_step_targets(CS, cp, want, np.float32(0.97))
        assert np.array_equal(traj.read(ra.TRAJ_TARGETS), td_o)
        v_o, _, _ = O.lanes_gae(CS, cp, want, 0.97, 0.95)
        assert np.array_equal(traj.read(ra.TRAJ_VALUES), v_o)
        # the optimisation steps against the oracle's loop on the same targets
        x, a = O.flat_samples(want)
        ad = L.oracle_adam_new(len(cp))
        ac = O.AdamCfg()
        L.oracle_adam_cfg_default(C.byref(ac))
        losses_o = np.zeros(steps, dtype=np.float32)
        c_o = cp.copy()
        L.oracle_critic_update_f32(CS, O.f32p(c_o), ad, C.byref(ac), O.f32p(x),
                                   O.f32p(np.ascontiguousarray(td_o.reshape(-1))), len(a), steps, O.f32p(losses_o))
        L.oracle_adam_free(ad)
        assert np.allclose(losses_d, losses_o, rtol=1e-4)
        assert np.abs(critic.get_params() - c_o).max() < 2e-5 + 1e-3 * steps * 1e-3
        assert st.steps == steps and losses_d[-1] < losses_d[0]
        # rl_critic_gradient is the MSE against RL_TRAJ_RETURNS whatever targets the last update regressed on
        # (the target pointer used to stay on the TD targets)
        c_now = critic.get_params()
        g_d, loss_d = ra.critic_gradient(critic, traj)
        _, _, rtg_o = O.lanes_gae(CS, cp, want, 0.97, 0.95)
        g_o = np.zeros_like(c_now)
        lo = C.c_float()
        L.oracle_critic_grad_f32(CS, O.f32p(c_now), O.f32p(x), O.f32p(np.ascontiguousarray(rtg_o.reshape(-1))), len(a),
                                 O.f32p(g_o), C.byref(lo))
        assert np.abs(g_d - g_o).max() <= 1e-5 * np.abs(g_o).max() and abs(loss_d - lo.value) <= 1e-5 * lo.value
    finally:
        engine.set_kernel_variant(0)


def test_reward_to_go_target_equals_the_returns_path(engine):
    """target = RewardToGo recomputes reward_to_go with the configured discount (critics/mod.rs:219-226): same numbers
    as rl_gae's returns, and the same update as rl_critic_update on them"""
    n, T, steps = 256, 48, 6
    traj, want = _cartpole(engine, n, T, max_steps=30)
    a, b = ra.Mlp(engine, 5, H, 1), ra.Mlp(engine, 5, H, 1)
    a.init(3)
    b.init(3)
    ra.gae(traj, a, 0.99, 0.95)
    rtg = traj.read(ra.TRAJ_RETURNS)
    _, la = ra.critic_update(a, ra.Adam(a), traj, steps, want_losses=True)
    cfg = ra.values_opt_config_default()
    cfg.opt_steps_per_update = steps
    assert (cfg.target, abs(cfg.discount_factor - 0.99) < 1e-7) == (ra.VALUE_TARGET_REWARD_TO_GO, True)
    _, lb = ra.values_opt_update(b, ra.Adam(b), traj, cfg, want_losses=True)
    assert np.array_equal(traj.read(ra.TRAJ_TARGETS), rtg)
    assert np.array_equal(la, lb) and np.array_equal(a.get_params(), b.get_params())
    # another discount: the targets follow the configuration, not what rl_gae left behind
    cfg.discount_factor = 0.9
    ra.values_opt_update(b, ra.Adam(b), traj, cfg)
    _, _, rtg9 = O.lanes_gae(CS, O.mlp_init(CS, 3), want, 0.9, 0.95)
    assert np.array_equal(traj.read(ra.TRAJ_TARGETS), rtg9)
    # and rl_critic_update afterwards is back on RL_TRAJ_RETURNS
    c = ra.Mlp(engine, 5, H, 1)
    c.init(3)
    _, lc = ra.critic_update(c, ra.Adam(c), traj, steps, want_losses=True)
    assert np.array_equal(la, lc)


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_one_step_td_recurrent_critic(engine, cell):
    n, T, steps, gamma = 64, 30, 3, np.float32(0.95)
    env = ra.ChainEnv(engine, n, max_steps=9, seed_env=3, seed_actor=4)
    make = ra.GruMlp if cell == "gru" else ra.LstmMlp
    pol, cri = make(engine, 5, 2), make(engine, 5, 1)
    pol.init(21)
    cri.init(22)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    want = traj.read_all()
    assert (want["flag"] == O.INTERRUPT).any()
    c0 = cri.get_params().copy()
    v_d, s_d = cri.seq_forward(traj)  # teacher-forced values / successor values (bit-exact vs the oracle elsewhere)
    cfg = ra.values_opt_config_default()
    cfg.opt_steps_per_update, cfg.target, cfg.discount_factor = steps, ra.VALUE_TARGET_ONE_STEP_TD, float(gamma)
    cst, losses_d = ra.values_opt_update(cri, ra.Adam(cri), traj, cfg, want_losses=True)
    td_o = O.seq_one_step_targets(v_d[0], s_d[0], want, gamma)
    td_d = traj.read(ra.TRAJ_TARGETS)
    assert np.array_equal(td_d, td_o)
    if cell == "gru":
        CSg = O.GruShape(5, 128, 128, 1)
        v_o, s_o = O.gru_seq_forward(CSg, c0, want, want_succ=True)
        assert np.array_equal(O.seq_one_step_targets(v_o[0], s_o[0], want, gamma), td_d)
        # the update loop on those targets
        B = n * T
        p = c0.copy()
        ost = L.oracle_adam_new(len(p))
        ocfg = O.AdamCfg()
        L.oracle_adam_cfg_default(C.byref(ocfg))
        losses_o = []
        for k in range(steps):
            v, _ = O.gru_seq_forward(CSg, p, want, want_succ=False)
            d = v[0] - td_o
            losses_o.append(float((d.astype(np.float64) ** 2).mean()))
            g = O.gru_seq_backward(CSg, p, want, (d * np.float32(2.0 / B))[None])
            L.oracle_adam_step_f32(ost, C.byref(ocfg), O.f32p(p), O.f32p(g))
        L.oracle_adam_free(ost)
        assert np.max(np.abs(losses_d - np.array(losses_o)) / np.array(losses_o)) < 2e-5
        assert np.mean(np.abs(cri.get_params() - p) < 3e-5) > 0.97
    assert losses_d[-1] < losses_d[0] and cst.steps == steps


def test_values_opt_argument_checks(engine):
    traj, _ = _cartpole(engine, 64, 8, max_steps=30)
    critic = ra.Mlp(engine, 5, H, 1)
    critic.init(1)
    with pytest.raises(ra.RelearnError):
        traj.read(ra.TRAJ_TARGETS)  # no targets before the first update
    cfg = ra.values_opt_config_default()
    cfg.target = 7
    with pytest.raises(ra.RelearnError) as e:
        ra.values_opt_update(critic, ra.Adam(critic), traj, cfg)
    assert e.value.code == ra.ERR_INVALID_ARGUMENT
