"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for discrete action indices, successor flags and everything on
the rollout path (both sides use include/rl_detmath.h + include/rl_chacha.h and explicit fma order);
stated fp32 tolerances for returns / advantages / gradients / parameters where the reduction order over
samples differs (device: blocked f32 partial sums; oracle: f64 accumulation rounded once).
"""
import ctypes as C

import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

H = 128
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)

# tolerances for sums over B samples in a different order (relative to the vector's max magnitude)
GRAD_RTOL = 1e-6
PARAM_ATOL = 2e-5


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(params=[0, 1], ids=["kernels-best", "kernels-v1"])
def variant(engine, request):
    """run the test once with the default (f32-MFMA fused) kernels and once with the v1 reference kernels"""
    engine.set_kernel_variant(request.param)
    yield request.param
    engine.set_kernel_variant(0)


def make_pair(engine, n, max_steps=500, limit=ra.LIMIT_VISIBLE, lane_offset=0, seed_env=0, seed_actor=1):
    env = ra.CartPoleEnv(engine, n, max_steps=max_steps, limit=limit, lane_offset=lane_offset, seed_env=seed_env,
                         seed_actor=seed_actor)
    sim = O.LaneSim(n, max_steps=max_steps, limit=limit, lane_offset=lane_offset, seed_env=seed_env,
                    seed_actor=seed_actor)
    return env, sim


def test_engine_is_gfx950(engine):
    name, arch, cus = engine.info()
    assert arch.startswith("gfx950")
    assert cus >= 200


def test_param_init_bit_exact(engine):
    for shape, seed in ((PS, 2), (CS, 3), (O.MlpShape(4, 64, 2), 11)):
        m = ra.Mlp(engine, shape.in_dim, shape.hidden, shape.out_dim)
        m.init(seed)
        assert np.array_equal(m.get_params(), O.mlp_init(shape, seed))
    # Glorot bound of Linear::new: U(+-sqrt(6 / (in + 1 + out)))
    p = O.mlp_init(PS, 2)
    assert np.abs(p[:5 * H + H]).max() <= np.float32(np.sqrt(6.0 / (5 + 1 + H)))
    assert np.abs(p[5 * H + H:]).max() <= np.float32(np.sqrt(6.0 / (H + 1 + 2)))


@pytest.mark.parametrize("limit,max_steps", [(ra.LIMIT_VISIBLE, 500), (ra.LIMIT_VISIBLE, 3), (ra.LIMIT_NONE, 0),
                                             (ra.LIMIT_LATENT, 7)])
def test_env_reset_observe_bit_exact(engine, limit, max_steps):
    env, sim = make_pair(engine, 1000, max_steps=max_steps or 1, limit=limit, lane_offset=12345, seed_env=7)
    st_d, st_o = env.get_state(), sim.get_state()
    for a, b in zip(st_d, st_o):
        assert np.array_equal(a, b)
    assert np.array_equal(env.observe(), sim.observe())
    assert np.all(np.abs(st_d[0]) <= 0.05)
    env.reset()
    sim.reset()
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)


def test_env_step_bit_exact_with_resets(engine):
    n = 2048
    env, sim = make_pair(engine, n, max_steps=9, seed_env=3)
    rng = np.random.default_rng(0)
    n_term = n_int = 0
    for t in range(60):
        actions = rng.integers(0, 2, size=n).astype(np.uint8)
        if t % 7 == 0:
            actions[:] = 1  # push one way to force pole falls
        r_d, f_d, o_d, t_d = env.step(actions)
        r_o, f_o, o_o, t_o = sim.step(actions)
        assert np.array_equal(f_d, f_o)
        assert np.array_equal(r_d, r_o)
        assert np.array_equal(o_d, o_o)
        m = f_o == O.INTERRUPT
        assert np.array_equal(t_d[:, m], t_o[:, m])
        n_term += int((f_o == O.TERMINATE).sum())
        n_int += int(m.sum())
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)
    assert n_int > 0, "the test must exercise step-limit interrupts"


def test_env_step_edge_states(engine):
    """friction-sign flip, both termination edges, -0.0 velocity, and a far-out-of-range angle"""
    n = 64
    env, sim = make_pair(engine, n, max_steps=500)
    st, nv, rem, rc = sim.get_state()
    st[:] = 0.0
    st[0, 0], st[1, 0] = 2.399, 3.0          # crosses +max_pos
    st[0, 1], st[1, 1] = -2.399, -3.0        # crosses -max_pos
    st[2, 2], st[3, 2] = 0.2094, 1.0         # crosses +12 degrees
    st[2, 3], st[3, 3] = -0.2094, -1.0
    st[1, 4] = -0.0                           # is_sign_positive(-0.0 * N) edge
    st[1, 5], nv[5] = -1.0, 1                 # cached sign inconsistent -> recompute branch
    st[1, 6], nv[6] = 1.0, 0
    st[2, 7] = 2.0                            # outside the observation space; dynamics must still agree
    st[2, 8] = -7.5
    st[3, 9] = 25.0                           # large angular velocity: normal force can change sign
    rem[10] = 1                               # interrupt on this very step
    for i in range(11, n):
        st[:, i] = np.random.default_rng(i).uniform(-0.2, 0.2, size=4)
        nv[i] = i % 2
    env.set_state(st, nv, rem, rc)
    sim.set_state(st, nv, rem, rc)
    for actions in (np.zeros(n, np.uint8), np.ones(n, np.uint8)):
        env.set_state(st, nv, rem, rc)
        sim.set_state(st, nv, rem, rc)
        r_d, f_d, o_d, t_d = env.step(actions)
        r_o, f_o, o_o, t_o = sim.step(actions)
        assert np.array_equal(f_d, f_o)
        assert np.array_equal(o_d, o_o)
        for a, b in zip(env.get_state(), sim.get_state()):
            assert np.array_equal(a, b)
        assert f_o[10] == O.INTERRUPT and t_d[4, 10] == 0.0
    assert f_o[0] == O.TERMINATE and f_o[2] == O.TERMINATE


@pytest.mark.parametrize("n,T,max_steps,lane_offset", [(256, 64, 500, 0), (1000, 37, 11, 4096), (64, 130, 40, 7)])
def test_rollout_bit_exact(engine, n, T, max_steps, lane_offset):
    env, sim = make_pair(engine, n, max_steps=max_steps, lane_offset=lane_offset, seed_env=5, seed_actor=6)
    policy = ra.Mlp(engine, 5, H, 2)
    policy.init(2)
    pp = policy.get_params()
    traj = ra.Trajectory(engine, n, T, 5)
    for period in range(2):  # lanes persist across periods: second period continues the episodes
        ra.rollout(env, policy, traj)
        got = traj.read_all()
        want = sim.rollout(PS, pp, T)
        assert np.array_equal(got["action"], want["action"])
        assert np.array_equal(got["flag"], want["flag"])
        assert np.array_equal(got["reward"], want["reward"])
        assert np.array_equal(got["obs"], want["obs"])
        m = want["flag"] == O.INTERRUPT
        assert np.array_equal(got["term_obs"][:, m], want["term_obs"][:, m])
        for a, b in zip(env.get_state(), sim.get_state()):
            assert np.array_equal(a, b)
    assert 0.2 < want["action"].mean() < 0.8


def test_rollout_sharding_invariance(engine):
    """lanes [k, k+m) of a big engine == an engine created with lane_offset = k (multi-GPU sharding rule)"""
    n, T = 512, 48
    policy = ra.Mlp(engine, 5, H, 2)
    policy.init(2)
    full_env = ra.CartPoleEnv(engine, n, max_steps=30)
    full = ra.Trajectory(engine, n, T, 5)
    ra.rollout(full_env, policy, full)
    f = full.read_all()
    half_env = ra.CartPoleEnv(engine, n // 2, max_steps=30, lane_offset=n // 2)
    half = ra.Trajectory(engine, n // 2, T, 5)
    ra.rollout(half_env, policy, half)
    h = half.read_all()
    for k in ("action", "flag", "obs"):
        assert np.array_equal(f[k][..., n // 2:], h[k])


def _traj_pair(engine, n, T, max_steps, seed=2):
    env, sim = make_pair(engine, n, max_steps=max_steps)
    policy = ra.Mlp(engine, 5, H, 2)
    policy.init(seed)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, policy, traj)
    want = sim.rollout(PS, policy.get_params(), T)
    return policy, traj, want


def test_values_and_gae_bit_exact(engine):
    n, T = 384, 96
    policy, traj, want = _traj_pair(engine, n, T, max_steps=25)
    critic = ra.Mlp(engine, 5, H, 1)
    critic.init(3)
    ra.gae(traj, critic, 0.99, 0.95)
    v_o, adv_o, rtg_o = O.lanes_gae(CS, critic.get_params(), want, 0.99, 0.95)
    assert np.array_equal(traj.read(ra.TRAJ_VALUES), v_o)
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), adv_o)
    assert np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)
    assert (want["flag"] == O.INTERRUPT).any() and (want["flag"] == O.TERMINATE).any()


def test_gae_linearity_and_gamma_one(engine):
    """size-independent properties: scaling rewards and the critic output by 2 scales advantages exactly by 2;
    with gamma = 1 and a zero critic the return is the number of steps left in the episode segment"""
    n, T = 256, 64
    policy, traj, want = _traj_pair(engine, n, T, max_steps=17)
    critic = ra.Mlp(engine, 5, H, 1)
    zero = np.zeros(critic.P, dtype=np.float32)
    critic.set_params(zero)
    ra.gae(traj, critic, 1.0, 1.0)
    rtg = traj.read(ra.TRAJ_RETURNS)
    flag = want["flag"]
    expect = np.zeros_like(rtg)
    run = np.zeros(n)
    for t in range(T - 1, -1, -1):
        ends = (flag[t] != 0) | (t == T - 1)
        run = np.where(ends, 1.0, run + 1.0)
        expect[t] = run
    assert np.array_equal(rtg, expect)
    adv = traj.read(ra.TRAJ_ADVANTAGES)
    assert np.array_equal(adv, expect)  # V == 0 => delta = r, lambda*gamma = 1
    cp = O.mlp_init(CS, 3)
    critic.set_params(cp)
    ra.gae(traj, critic, 0.99, 0.95)
    a1 = traj.read(ra.TRAJ_ADVANTAGES)
    cp2 = cp.copy()
    cp2[5 * H + H:] *= 2.0  # doubling the output layer doubles V exactly
    critic.set_params(cp2)
    traj.write(ra.TRAJ_REWARD, want["reward"] * 2.0)
    ra.gae(traj, critic, 0.99, 0.95)
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), a1 * 2.0)


def _with_advantages(engine, n, T, max_steps=30):
    policy, traj, want = _traj_pair(engine, n, T, max_steps)
    critic = ra.Mlp(engine, 5, H, 1)
    critic.init(3)
    ra.gae(traj, critic, 0.99, 0.95)
    adv = traj.read(ra.TRAJ_ADVANTAGES)
    rtg = traj.read(ra.TRAJ_RETURNS)
    x, a = O.flat_samples(want)
    return policy, critic, traj, x, a, np.ascontiguousarray(adv.reshape(-1)), np.ascontiguousarray(rtg.reshape(-1))


def test_policy_gradient_matches_oracle(engine, variant):
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, 512, 64)
    g_d, loss_d, ent_d = ra.policy_gradient(policy, traj)
    pp = policy.get_params()
    g_o = np.zeros_like(pp)
    loss_o = C.c_float()
    O.lib().oracle_policy_grad_f32(PS, O.f32p(pp), O.f32p(x), O.i64p(a), O.f32p(adv), len(a), O.f32p(g_o),
                                   C.byref(loss_o))
    assert rel_err(g_d, g_o) < GRAD_RTOL
    assert abs(loss_d - loss_o.value) <= 1e-5 * max(1.0, abs(loss_o.value))
    # f64 ground truth: the f32 device result must be as close to it as the f32 oracle is (x4 slack)
    g64 = np.zeros(len(pp), dtype=np.float64)
    l64 = C.c_double()
    O.lib().oracle_policy_grad_f64(PS, O.f64p(pp.astype(np.float64)), O.f64p(x.astype(np.float64)), O.i64p(a),
                                   O.f64p(adv.astype(np.float64)), len(a), O.f64p(g64), C.byref(l64))
    assert rel_err(g_d, g64) < max(4 * rel_err(g_o, g64), 1e-5)
    assert 0.0 < ent_d <= np.log(2.0) + 1e-6


def test_policy_fvp_matches_oracle_and_is_symmetric(engine, variant):
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, 512, 64)
    pp = policy.get_params()
    rng = np.random.default_rng(1)
    u = rng.standard_normal(policy.P).astype(np.float32)
    v = rng.standard_normal(policy.P).astype(np.float32)
    hv_d = ra.policy_fvp(policy, traj, v, 1e-5)
    hv_o = np.zeros_like(pp)
    O.lib().oracle_policy_fvp_f32(PS, O.f32p(pp), O.f32p(x), len(a), O.f32p(v), 1e-5, O.f32p(hv_o))
    assert rel_err(hv_d, hv_o) < GRAD_RTOL
    hu_d = ra.policy_fvp(policy, traj, u, 1e-5)
    # symmetry u^T H v == v^T H u and positive semi-definiteness v^T H v >= 0
    lhs, rhs = float(np.dot(u.astype(np.float64), hv_d)), float(np.dot(v.astype(np.float64), hu_d))
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs), 1e-6)
    assert float(np.dot(v.astype(np.float64), hv_d)) > 0.0


def test_policy_loss_kl_matches_oracle(engine, variant):
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, 512, 64)
    p0 = policy.get_params()
    rng = np.random.default_rng(2)
    p1 = (p0 + 0.01 * rng.standard_normal(len(p0))).astype(np.float32)
    policy.set_params(p1)
    loss_d, kl_d = ra.policy_loss_kl(policy, traj, p0)
    lo, ko = C.c_float(), C.c_float()
    O.lib().oracle_policy_loss_kl_f32(PS, O.f32p(p1), O.f32p(p0), O.f32p(x), O.i64p(a), O.f32p(adv), len(a),
                                      C.byref(lo), C.byref(ko))
    assert abs(loss_d - lo.value) <= 1e-5 * max(1.0, abs(lo.value))
    assert abs(kl_d - ko.value) <= 1e-5 * max(1e-3, abs(ko.value))
    # KL(theta0 || theta0) == 0 exactly and loss == -mean(A)
    policy.set_params(p0)
    loss0, kl0 = ra.policy_loss_kl(policy, traj, p0)
    assert kl0 == 0.0
    assert abs(loss0 + adv.astype(np.float64).mean()) < 1e-5 * max(1.0, abs(adv.mean()))


def _oracle_cfg(dcfg):
    cfg = O.TrpoCfg()
    O.lib().oracle_trpo_cfg_default(C.byref(cfg))
    cfg.iterations, cfg.max_backtracks = dcfg.iterations, dcfg.max_backtracks
    cfg.backtrack_ratio, cfg.hpv_reg_coeff = dcfg.backtrack_ratio, dcfg.hpv_reg_coeff
    cfg.max_kl, cfg.accept_violation = dcfg.max_policy_step_kl, dcfg.accept_violation
    return cfg


@pytest.mark.parametrize("iterations,tol", [(1, 3e-4), (2, 2e-3)])
def test_trpo_update_few_cg_iterations_tight(engine, variant, iterations, tol):
    """With 1-2 CG iterations rounding is not amplified: the whole pipeline (gradient, Fisher-vector products,
    CG bookkeeping, step size, line search, acceptance) must agree with the f32 oracle tightly."""
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, 512, 64, 500)
    p0 = policy.get_params()
    dcfg = ra.trpo_config_default()
    dcfg.iterations = iterations
    st_d = ra.trpo_update(policy, traj, dcfg)
    p_d = policy.get_params()
    p_o, st_o, sd_o = O.trpo_update(PS, p0, x, a, adv, _oracle_cfg(dcfg))
    assert st_d.status == st_o.status
    assert st_d.num_backtracks == st_o.num_backtracks
    assert st_d.cg_iterations == st_o.cg_iterations == iterations
    assert abs(st_d.entropy - st_o.entropy) < 1e-5
    assert abs(st_d.loss_initial - st_o.loss_initial) <= 1e-5 * max(1.0, abs(st_o.loss_initial))
    assert abs(st_d.step_size - st_o.step_size) <= tol * st_o.step_size
    assert abs(st_d.loss_final - st_o.loss_final) <= 1e-5 * max(1.0, abs(st_o.loss_final))
    assert abs(st_d.constraint_val_final - st_o.constraint_val_final) <= 10 * tol * st_o.constraint_val_final + 1e-7
    assert np.abs(p_d - p_o).max() <= tol * np.abs(p_o - p0).max() + 1e-7


@pytest.mark.parametrize("n,T,max_steps", [(256, 32, 30), (1024, 128, 500)])
def test_trpo_update_default_config_vs_f64_truth(engine, variant, n, T, max_steps):
    """10 CG iterations in f32 on the Fisher matrix of this MLP are ill-conditioned: two correct f32
    implementations differ from each other by tens of percent in the step direction (the f32 and f64 oracles
    differ by 20-40 % here).  Stated tolerance: the device result must be no farther from the f64 ground truth
    than twice the f32 restatement is, and must satisfy the trust-region acceptance rule itself."""
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, n, T, max_steps)
    p0 = policy.get_params()
    st_d = ra.trpo_update(policy, traj)
    p_d = policy.get_params()
    p32, st32, sd32 = O.trpo_update(PS, p0, x, a, adv)
    p64, st64, sd64 = O.trpo_update(PS, p0, x, a, adv, f64=True)
    assert st_d.status == st32.status == st64.status == ra.OPT_OK
    assert st_d.cg_iterations == st32.cg_iterations == st64.cg_iterations
    assert abs(st_d.entropy - st64.entropy) < 1e-5
    assert abs(st_d.loss_initial - st64.loss_initial) <= 1e-5 * max(1.0, abs(st64.loss_initial))
    err_dev = abs(st_d.step_size - st64.step_size)
    err_o32 = abs(st32.step_size - st64.step_size)
    assert err_dev <= 2.0 * err_o32 + 1e-3 * st64.step_size
    lo = min(st32.num_backtracks, st64.num_backtracks) - 1
    hi = max(st32.num_backtracks, st64.num_backtracks) + 1
    assert lo <= st_d.num_backtracks <= hi
    # the accepted step obeys the trust region and improves the surrogate (conjugate_gradient.rs:218)
    assert st_d.constraint_val_final <= 0.01 and st_d.loss_final < st_d.loss_initial
    assert np.abs(p_d - p0).max() > 0
    # and what was accepted is what the device says: re-evaluate loss / KL of the new parameters independently
    lo_, ko_ = C.c_float(), C.c_float()
    O.lib().oracle_policy_loss_kl_f32(PS, O.f32p(p_d), O.f32p(p0), O.f32p(x), O.i64p(a), O.f32p(adv), len(a),
                                      C.byref(lo_), C.byref(ko_))
    assert abs(lo_.value - st_d.loss_final) <= 1e-5 * max(1.0, abs(lo_.value))
    assert abs(ko_.value - st_d.constraint_val_final) <= 1e-4 * ko_.value + 1e-8
    # A second angle on the same step: the BACKWARD error of the step direction, |(F + reg I) x
    # - g| / |g| with the f64 operator and gradient of the oracle as the yardstick.  Ten CG iterations do not solve the
    # system — the f64 run itself stops at 0.038 / 0.0075 on the two cases — and in f32 the recurrence loses
    # conjugacy: the f32 ORACLE's direction has 0.32 / 0.040, the device's 0.25 / 0.026 (fused kernels) and 0.046 / 0.11
    # (v1 kernels).  No f32 evaluation is closer than a factor of a few to another here, in the residual as in the step
    # itself; the bar is 3 x the worse of the two oracles, which a wrong operator, a wrong sign or a skipped iteration
    # (residual of order 1) cannot meet.
    g64, _ = O.grad_f64_mt("policy", PS, p0, x, a.astype(np.uint8), adv)

    def residual(direction):
        d = np.asarray(direction, dtype=np.float64)
        fx, _ = O.grad_f64_mt("fvp", PS, p0, x, v=d.astype(np.float32))
        r = fx + 1e-5 * d - g64
        return np.linalg.norm(r) / np.linalg.norm(g64)

    x_dev = (p0.astype(np.float64) - p_d.astype(np.float64)) / (st_d.step_scale * st_d.step_size)
    r_dev, r_o32, r_o64 = residual(x_dev), residual(sd32), residual(sd64)
    print("TRPO backward error |Ax - g| / |g|: device %.4g, f32 oracle %.4g, f64 oracle %.4g" % (r_dev, r_o32, r_o64))
    assert r_dev <= 3.0 * max(r_o32, r_o64) + 1e-6


def test_trpo_rollback_on_failure(engine, variant):
    """Line-search failures restore the parameters and report the reference's error kinds
    (conjugate_gradient.rs:228-253); the device must classify exactly like the oracle."""
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, 256, 32)
    p0 = policy.get_params()
    # (a) budget too small: this short-episode batch needs ~13 backtracks, allow 3
    cfg = ra.trpo_config_default()
    cfg.max_backtracks = 3
    st = ra.trpo_update(policy, traj, cfg)
    p_o, st_o, _ = O.trpo_update(PS, p0, x, a, adv, _oracle_cfg(cfg))
    assert st.status == st_o.status
    assert st.status in (ra.OPT_CONSTRAINT_VIOLATED, ra.OPT_LOSS_NOT_IMPROVING)
    assert st.num_backtracks == -1 and st_o.num_backtracks == -1
    assert np.array_equal(policy.get_params(), p0) and np.array_equal(p_o, p0)
    # (b) no CG iterations: zero step, loss cannot improve
    cfg = ra.trpo_config_default()
    cfg.iterations = 0
    st = ra.trpo_update(policy, traj, cfg)
    assert st.status == ra.OPT_LOSS_NOT_IMPROVING and st.cg_iterations == 0
    assert np.array_equal(policy.get_params(), p0)
    # (c) zero advantages: g = 0 -> alpha = 0/0 = NaN -> nan_to_num(step_dir) = 0 (conjugate_gradient.rs:152)
    traj.write(ra.TRAJ_ADVANTAGES, np.zeros((32, 256), np.float32))
    st = ra.trpo_update(policy, traj)
    p_o, st_o, _ = O.trpo_update(PS, p0, x, a, np.zeros_like(adv))
    assert st.status == st_o.status == ra.OPT_LOSS_NOT_IMPROVING
    assert np.array_equal(policy.get_params(), p0)


def test_critic_gradient_and_update_match_oracle(engine, variant):
    policy, critic, traj, x, a, adv, rtg = _with_advantages(engine, 512, 64)
    cp = critic.get_params()
    g_d, loss_d = ra.critic_gradient(critic, traj)
    g_o = np.zeros_like(cp)
    lo = C.c_float()
    O.lib().oracle_critic_grad_f32(CS, O.f32p(cp), O.f32p(x), O.f32p(rtg), len(a), O.f32p(g_o), C.byref(lo))
    assert rel_err(g_d, g_o) < GRAD_RTOL
    assert abs(loss_d - lo.value) <= 1e-5 * lo.value
    steps = 20
    opt = ra.Adam(critic)
    st, losses_d = ra.critic_update(critic, opt, traj, steps, want_losses=True)
    ad = O.lib().oracle_adam_new(len(cp))
    ac = O.AdamCfg()
    O.lib().oracle_adam_cfg_default(C.byref(ac))
    losses_o = np.zeros(steps, dtype=np.float32)
    c_o = cp.copy()
    O.lib().oracle_critic_update_f32(CS, O.f32p(c_o), ad, C.byref(ac), O.f32p(x), O.f32p(rtg), len(a), steps,
                                     O.f32p(losses_o))
    O.lib().oracle_adam_free(ad)
    assert np.allclose(losses_d, losses_o, rtol=1e-4)
    assert np.abs(critic.get_params() - c_o).max() < PARAM_ATOL + 1e-3 * steps * 1e-3
    assert losses_d[-1] < losses_d[0]
    assert st.steps == steps


@pytest.mark.parametrize("n,T,target", [(512, 64, ra.VALUE_TARGET_REWARD_TO_GO), (8192, 32, ra.VALUE_TARGET_REWARD_TO_GO),
                                        (1024, 48, ra.VALUE_TARGET_ONE_STEP_TD)])
def test_actor_critic_update_equals_the_two_updates_in_turn(engine, n, T, target):
    """rl_actor_critic_update runs the TRPO chain and the critic chain side by side on two streams; each chain's
    arithmetic and order are those of rl_trpo_update / rl_values_opt_update, so every number it produces — parameters,
    optimiser moments (through a further step), statistics, the per-step losses — is BIT-identical to calling the two
    one after the other (policy.update then critic.update, actor_critic.rs:196-208), over two periods, and identical
    again with the chains forced onto one stream."""
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update, ccfg.target = 12, target

    def run(mode):
        env = ra.CartPoleEnv(engine, n, max_steps=60, seed_env=21, seed_actor=22)
        pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
        pol.init(2)
        cri.init(3)
        opt = ra.Adam(cri)
        traj = ra.Trajectory(engine, n, T, 5)
        out = []
        engine.set_serial_update(mode == "serial")
        try:
            for period in range(2):
                ra.rollout(env, pol, traj)
                ra.gae(traj, cri, 0.99, 0.95)
                if mode == "separate":
                    pst = ra.trpo_update(pol, traj)
                    cst, losses = ra.values_opt_update(cri, opt, traj, ccfg, want_losses=True)
                else:
                    pst, cst, losses = ra.actor_critic_update(pol, cri, opt, traj, None, ccfg, want_losses=True)
                out.append((pol.get_params(), cri.get_params(), pst.as_dict(), (cst.loss_first, cst.loss_last, cst.steps),
                            losses.copy(), traj.read(ra.TRAJ_TARGETS) if target == ra.VALUE_TARGET_ONE_STEP_TD else None))
        finally:
            engine.set_serial_update(False)
        return out

    ref = run("separate")
    for mode in ("overlap", "serial"):
        got = run(mode)
        for (p0, c0, s0, k0, l0, t0), (p1, c1, s1, k1, l1, t1) in zip(ref, got):
            assert np.array_equal(p0, p1) and np.array_equal(c0, c1), mode
            assert s0 == s1 and k0 == k1 and np.array_equal(l0, l1), mode
            assert t0 is None or np.array_equal(t0, t1)
    assert ref[0][2]["status"] == ra.OPT_OK and ref[0][4][-1] < ref[0][4][0]


def _drop_pending_update(engine):
    """a failed test must not leave the session's engine with an update pending (the trajectory it belongs to is
    destroyed with the test's locals, which ends it; a traceback may keep those alive, so end it by hand)"""
    engine.sync()
    for t in _LIVE_TRAJS:
        if t.h:
            ra.lib().rl_actor_critic_update_finish(t.h, None, None)  # (an error code when nothing is pending on it)
    del _LIVE_TRAJS[:]


_LIVE_TRAJS = []


@pytest.mark.parametrize("n,T,target", [(512, 64, ra.VALUE_TARGET_REWARD_TO_GO), (8192, 32, ra.VALUE_TARGET_REWARD_TO_GO),
                                        (1024, 48, ra.VALUE_TARGET_ONE_STEP_TD)])
def test_pipelined_periods_equal_the_serial_sequence(engine, n, T, target):
    """rl_actor_critic_update_begin / _finish: period k + 1's rollout (into the other trajectory of a pair) is enqueued
    as soon as TRPO k has finished, under critic chain k; rl_gae of k + 1 orders itself behind that chain on the device.
    The next collection reads only the updated policy (agents/mod.rs:48-59; actor_critic.rs:196-208 orders policy.update
    before critic.update), so every number over three periods — trajectories, advantages, parameters, statistics,
    per-step critic losses — is BIT-identical to rollout / rl_gae / rl_trpo_update / rl_values_opt_update in turn."""
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update, ccfg.target = 12, target
    periods = 3

    def snapshot(traj, pol, cri, pst, cst, losses):
        return (traj.read_all(), traj.read(ra.TRAJ_ADVANTAGES), pol.get_params(), cri.get_params(), pst.as_dict(),
                (cst.loss_first, cst.loss_last, cst.steps), losses.copy())

    def run(mode):
        env = ra.CartPoleEnv(engine, n, max_steps=60, seed_env=21, seed_actor=22)
        pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
        pol.init(2)
        cri.init(3)
        opt = ra.Adam(cri)
        trajs = [ra.Trajectory(engine, n, T, 5), ra.Trajectory(engine, n, T, 5)]
        _LIVE_TRAJS.extend(trajs)
        out = []
        if mode == "serial":
            for k in range(periods):
                tr = trajs[k & 1]
                ra.rollout(env, pol, tr)
                ra.gae(tr, cri, 0.99, 0.95)
                pst = ra.trpo_update(pol, tr)
                cst, losses = ra.values_opt_update(cri, opt, tr, ccfg, want_losses=True)
                out.append(snapshot(tr, pol, cri, pst, cst, losses))
            return out
        # pipelined: nothing is read back between _begin(k) and the rollout of k + 1 (a read-back would order itself
        # behind the critic chain: still correct, but no longer the overlapped sequence this test is about)
        stats = []
        for k in range(periods):
            tr = trajs[k & 1]
            ra.rollout(env, pol, tr)                      # beside critic chain k - 1 (other trajectory)
            ra.gae(tr, cri, 0.99, 0.95)                   # device-side wait for critic chain k - 1
            if k > 0:
                cst, losses = ra.actor_critic_update_finish(trajs[(k - 1) & 1], want_losses=True)
                stats[-1] += (cst, losses, cri.get_params(), trajs[(k - 1) & 1].read_all(),
                              trajs[(k - 1) & 1].read(ra.TRAJ_ADVANTAGES))
            pst = ra.actor_critic_update_begin(pol, cri, opt, tr, None, ccfg)
            # (reading the policy waits for the main stream only after a settle: do it through the next period's reads)
            stats.append((pst,))
        cst, losses = ra.actor_critic_update_finish(trajs[(periods - 1) & 1], want_losses=True)
        stats[-1] += (cst, losses, cri.get_params(), trajs[(periods - 1) & 1].read_all(),
                      trajs[(periods - 1) & 1].read(ra.TRAJ_ADVANTAGES))
        return stats, pol.get_params()

    ref = run("serial")
    try:
        got, pol_final = run("pipelined")
    finally:
        _drop_pending_update(engine)
    for k in range(periods):
        tr0, adv0, p0, c0, s0, k0, l0 = ref[k]
        pst, cst, losses, c1, tr1, adv1 = got[k]
        for f in ("action", "flag", "reward", "obs"):
            assert np.array_equal(tr0[f], tr1[f]), (k, f)
        assert np.array_equal(adv0, adv1), k
        assert s0 == pst.as_dict(), k
        assert k0 == (cst.loss_first, cst.loss_last, cst.steps) and np.array_equal(l0, losses), k
        # the critic read after _finish(k) of the pipelined run has seen exactly chain k
        assert np.array_equal(c0, c1), k
    assert np.array_equal(ref[-1][2], pol_final)
    assert ref[0][4]["status"] == ra.OPT_OK


def test_pending_update_orders_every_other_call_behind_the_critic_chain(engine):
    """between _begin and _finish: reads of the critic, a rollout into the SAME trajectory and a second _begin — the first
    two are ordered behind the chain (they see the finished critic / do not corrupt the chain's inputs), the third is an
    error; _finish without _begin is an error too"""
    n, T = 2048, 32
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update = 20

    def setup():
        env = ra.CartPoleEnv(engine, n, max_steps=60, seed_env=5, seed_actor=6)
        pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
        pol.init(2)
        cri.init(3)
        _LIVE_TRAJS.append(ra.Trajectory(engine, n, T, 5))
        return env, pol, cri, ra.Adam(cri), _LIVE_TRAJS[-1]

    env, pol, cri, opt, traj = setup()
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    _, cst_ref, losses_ref = ra.actor_critic_update(pol, cri, opt, traj, None, ccfg, want_losses=True)
    cri_ref = cri.get_params()

    env, pol, cri, opt, traj = setup()
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    with pytest.raises(ra.RelearnError):
        ra.actor_critic_update_finish(traj)
    ra.actor_critic_update_begin(pol, cri, opt, traj, None, ccfg)
    try:
        with pytest.raises(ra.RelearnError):
            ra.actor_critic_update_begin(pol, cri, opt, traj, None, ccfg)
        assert np.array_equal(cri.get_params(), cri_ref)  # ordered behind the chain
        ra.rollout(env, pol, traj)                    # same trajectory: after the chain, so the chain read the old one
        cst, losses = ra.actor_critic_update_finish(traj, want_losses=True)
    finally:
        _drop_pending_update(engine)
    assert np.array_equal(losses, losses_ref) and cst.loss_last == cst_ref.loss_last
    with pytest.raises(ra.RelearnError):
        ra.actor_critic_update_finish(traj)


def test_adam_step_matches_oracle(engine):
    m = ra.Mlp(engine, 5, H, 1)
    m.init(9)
    p = m.get_params()
    opt = ra.Adam(m)
    ad = O.lib().oracle_adam_new(len(p))
    ac = O.AdamCfg()
    O.lib().oracle_adam_cfg_default(C.byref(ac))
    rng = np.random.default_rng(3)
    p_o = p.copy()
    for k in range(5):
        g = (rng.standard_normal(len(p)) * 10.0 ** rng.integers(-6, 2)).astype(np.float32)
        opt.step_host(g)
        O.lib().oracle_adam_step_f32(ad, C.byref(ac), O.f32p(p_o), O.f32p(g))
    O.lib().oracle_adam_free(ad)
    # identical formulas; only pow()/sqrt() of the f64 bias corrections may differ in the last place
    assert np.abs(m.get_params() - p_o).max() <= 2e-7


def test_error_paths(engine):
    with pytest.raises(ra.RelearnError) as e:
        ra.Mlp(engine, 7, 128, 2)
    assert e.value.code == ra.ERR_BUILD_AGENT
    with pytest.raises(ra.RelearnError) as e:
        ra.CartPoleEnv(engine, 16, max_steps=0)
    assert e.value.code == ra.ERR_BUILD_ENV
    env = ra.CartPoleEnv(engine, 16)
    traj = ra.Trajectory(engine, 32, 8, 5)
    pol = ra.Mlp(engine, 5, 16, 2)
    with pytest.raises(ra.RelearnError) as e:
        ra.rollout(env, pol, traj)
    assert e.value.code == ra.ERR_INVALID_ARGUMENT
    buf = np.zeros(4, np.uint8)
    with pytest.raises(ra.RelearnError) as e:
        ra._check(ra.lib().rl_traj_read(traj.h, C.c_int32(99), buf.ctypes.data_as(C.c_void_p), C.c_uint64(4)),
                  engine.h)
    assert e.value.code == ra.ERR_INVALID_ARGUMENT
    assert b"unknown trajectory field" in ra.lib().rl_last_error(engine.h)


def test_rccl_call_path_single_rank():
    """dlopen of librccl, ncclGetUniqueId, ncclCommInitRank and ncclAllReduce(sum, f32) on the engine's own
    stream, exercised with a 1-rank communicator (RELEARN_FORCE_RCCL=1): results must equal the no-communicator
    run bit for bit (an all-reduce over one rank is the identity)."""
    import os
    uid = ra.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    results = []
    for force in (False, True):
        if force:
            os.environ["RELEARN_FORCE_RCCL"] = "1"
        try:
            eng = ra.Engine(0)
            if force:
                eng.comm_init(0, 1, ra.comm_unique_id())
            policy, critic, traj, x, a, adv, rtg = _with_advantages(eng, 256, 32)
            st = ra.trpo_update(policy, traj)
            opt = ra.Adam(critic)
            cst = ra.critic_update(critic, opt, traj, 5)
            results.append((policy.get_params(), critic.get_params(), st.step_size, cst.loss_last))
            eng.profile_enable(True)
            ra.critic_update(critic, opt, traj, 2)
            assert eng.profile_read(reset=True)["allreduce"][1] == (2 if force else 0)  # the collective really ran
            # the two chains side by side, each on its own communicator (rl_comm_init builds the second one through the
            # first): 3 critic all-reduces on the auxiliary stream next to the TRPO chain's on the main stream
            ccfg = ra.values_opt_config_default()
            ccfg.opt_steps_per_update = 3
            pst, cst2 = ra.actor_critic_update(policy, critic, opt, traj, None, ccfg)
            if force:
                assert eng.profile_read()["allreduce"][1] >= 3 + 1 + 11 + 1
            results[-1] = results[-1] + (policy.get_params(), critic.get_params(), pst.step_size, cst2.loss_last)
            for o in (opt, traj, critic, policy):
                o.close()
            eng.close()
        finally:
            os.environ.pop("RELEARN_FORCE_RCCL", None)
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2] and results[0][3] == results[1][3]
    assert np.array_equal(results[0][4], results[1][4]) and np.array_equal(results[0][5], results[1][5])
    assert results[0][6] == results[1][6] and results[0][7] == results[1][7]


# ---------------------------------------------------------------- round 6: the weight image and the shared return plane
def test_weight_image_kept_by_the_parameter_writers_equals_a_fresh_one(engine):
    """The fused kernels read their weight pieces from the module's weight image (relearn_amd/csrc/bf16_tile.hpp): built
    by the first fused launch of a C-ABI call, then kept current by the kernels that write the parameters (k_reduce_adam
    for the critic, k_ls_set_params for the line search's candidates).  A K-step critic update in ONE call (steps 2..K
    read the image k_reduce_adam left) must equal K one-step calls (every call rebuilds the image from the flat vector)
    bit for bit; and the loss / KL the line search accepted (evaluated on an image k_ls_set_params wrote) must be the
    numbers rl_policy_loss_kl computes for the same two parameter vectors from images built from scratch."""
    n, T, K = 2048, 32, 6
    env = ra.CartPoleEnv(engine, n, max_steps=60, seed_env=31, seed_actor=32)
    pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
    pol.init(2)
    cri.init(3)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    c0 = cri.get_params()
    opt = ra.Adam(cri)
    st, losses = ra.critic_update(cri, opt, traj, K, want_losses=True)
    one_call = cri.get_params()
    cri.set_params(c0)
    opt2 = ra.Adam(cri)
    step_losses = []
    for _ in range(K):
        _, l1 = ra.critic_update(cri, opt2, traj, 1, want_losses=True)
        step_losses.append(l1[0])
    assert np.array_equal(one_call, cri.get_params())
    assert np.array_equal(losses, np.asarray(step_losses, dtype=np.float32))
    p0 = pol.get_params()
    pst = ra.trpo_update(pol, traj)
    assert pst.status == ra.OPT_OK and pst.num_backtracks >= 0
    loss, kl = ra.policy_loss_kl(pol, traj, p0)
    assert np.float32(pst.loss_final) == np.float32(loss) and np.float32(pst.constraint_val_final) == np.float32(kl)


def test_reward_to_go_targets_come_from_the_advantage_scan_only_when_they_are_the_same_numbers(engine):
    """RewardToGo value targets (critics/mod.rs:101-105, 203-229) at the discount factor of the advantage pass are the
    return plane rl_gae has just written — the critic update regresses on it without a second scan; at another discount
    factor, or after the rewards were rewritten, they are scanned afresh.  Either way they equal the oracle's."""
    n, T = 512, 40
    env = ra.CartPoleEnv(engine, n, max_steps=25, seed_env=41, seed_actor=42)
    pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
    pol.init(2)
    cri.init(3)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.97, 0.95)
    tr = traj.read_all()

    def oracle_rtg(reward, flag, gamma):
        out = np.zeros_like(reward)
        nxt = np.zeros(reward.shape[1], dtype=np.float32)
        for t in range(reward.shape[0] - 1, -1, -1):
            ends = (flag[t] != ra.SUCC_CONTINUE) | (t == reward.shape[0] - 1)
            g = np.where(ends, reward[t], reward[t] + (nxt * np.float32(gamma)).astype(np.float32)).astype(np.float32)
            out[t] = g
            nxt = g
        return out

    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update = 1
    for gamma in (0.97, 0.99):  # the scan's own factor (shared plane), then another one (own scan)
        ccfg.discount_factor = gamma
        ra.values_opt_update(cri, ra.Adam(cri), traj, ccfg)
        assert np.array_equal(traj.read(ra.TRAJ_TARGETS), oracle_rtg(tr["reward"], tr["flag"], gamma)), gamma
    assert np.array_equal(traj.read(ra.TRAJ_RETURNS), oracle_rtg(tr["reward"], tr["flag"], 0.97))
    # rewards rewritten by the host: the return plane no longer vouches for anything
    reward2 = (tr["reward"] * np.float32(0.5)).astype(np.float32)
    traj.write(ra.TRAJ_REWARD, reward2)
    ccfg.discount_factor = 0.97
    ra.values_opt_update(cri, ra.Adam(cri), traj, ccfg)
    assert np.array_equal(traj.read(ra.TRAJ_TARGETS), oracle_rtg(reward2, tr["flag"], 0.97))
