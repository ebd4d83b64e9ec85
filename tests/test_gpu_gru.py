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


# ------------------------------------------------------------------ gradients through time and the update loops
GRAD_RTOL = 5e-6   # device f32 (MFMA partial sums over <= 2048 samples, then f64) vs the f64 evaluation, rel. to max|g|


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def setup_update(engine, n=64, T=30, max_steps=9, lr=1e-3):
    env, sim = chain_pair(engine, n, max_steps=max_steps)
    pol, cri = ra.GruMlp(engine, 5, 2), ra.GruMlp(engine, 5, 1)
    pol.init(21)
    cri.init(22)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    want = sim.rollout_gru(PS, pol.get_params(), T)
    ra.gae(traj, cri, 0.95, 0.9)
    want["adv"] = traj.read(ra.TRAJ_ADVANTAGES)
    want["rtg"] = traj.read(ra.TRAJ_RETURNS)
    acfg = ra.adam_config_default()
    acfg.learning_rate = lr
    ocfg = O.AdamCfg()
    L.oracle_adam_cfg_default(C.byref(ocfg))
    ocfg.lr = lr
    return pol, cri, traj, want, acfg, ocfg


def policy_grad_f64(p, want):
    """-mean(ratio * A) at ratio 1 through time, everything in f64 (numpy softmax + the f64 BPTT oracle)"""
    logits, _ = O.gru_seq_forward(PS, p, want, f64=True, want_succ=False)
    B = logits[0].size
    z = logits - logits.max(0)
    lp = z - np.log(np.exp(z).sum(0))
    pr = np.exp(lp)
    a = want["action"].astype(np.int64)
    ind = np.stack([a == 0, a == 1]).astype(np.float64)
    dl = -(want["adv"].astype(np.float64) / B) * (ind - pr)
    return O.gru_seq_backward(PS, p, want, dl, f64=True)


def test_policy_gradient_through_time(engine):
    pol, cri, traj, want, _, _ = setup_update(engine)
    p = pol.get_params()
    g_d, loss_d, ent_d = ra.policy_gradient(pol, traj)
    logits, _ = O.gru_seq_forward(PS, p, want, want_succ=False)
    dl, lp, loss_sum, ent_sum = O.seq_policy_dlogits(logits, want["action"], want["adv"])
    g32 = O.gru_seq_backward(PS, p, want, dl)
    g64 = policy_grad_f64(p, want)
    B = want["action"].size
    assert rel_err(g_d, g64) < GRAD_RTOL, (rel_err(g_d, g64), rel_err(g32, g64))
    assert abs(loss_d + loss_sum / B) <= 1e-5 * max(1.0, abs(loss_sum / B))
    assert abs(ent_d - ent_sum / B) < 1e-5
    # every parameter block receives gradient
    H, D = 128, 5
    cuts = np.cumsum([3 * H * D, 3 * H * H, 3 * H, 3 * H, H * H, H, 2 * H, 2])
    for lo, hi in zip(np.r_[0, cuts[:-1]], cuts):
        assert np.abs(g_d[lo:hi]).max() > 0
        assert rel_err(g_d[lo:hi], g64[lo:hi]) < 50 * GRAD_RTOL


def test_critic_gradient_through_time(engine):
    pol, cri, traj, want, _, _ = setup_update(engine)
    p = cri.get_params()
    g_d, loss_d = ra.critic_gradient(cri, traj)
    v, _ = O.gru_seq_forward(CS, p, want, f64=True, want_succ=False)
    B = v[0].size
    d = v - want["rtg"].astype(np.float64)[None]
    g64 = O.gru_seq_backward(CS, p, want, 2.0 * d / B, f64=True)
    assert rel_err(g_d, g64) < GRAD_RTOL
    assert abs(loss_d - (d * d).mean()) <= 1e-5 * (d * d).mean()


@pytest.mark.parametrize("n,T,max_steps", [(32, 1, 9), (32, 2, 1), (96, 7, 3), (160, 3, 2), (2048, 20, 4), (2048, 50, 7)],
                         ids=["one-tile-one-step", "every-step-ends", "three-tiles", "five-tiles", "two-blocks-per-chunk",
                              "steady-state-chunks"])
def test_gradients_through_time_at_edge_shapes(engine, n, T, max_steps):
    """shapes at the edges of the training kernels' loops: a single (step, tile) block, episodes that end at every
    step, tile counts that are not a power of two, and enough blocks (1,280 and 3,200) for the weight-gradient
    kernel's chunks to hold two blocks and to reach the branch-free steady state of its loop"""
    pol, cri, traj, want, _, _ = setup_update(engine, n=n, T=T, max_steps=max_steps)
    g_pol = ra.policy_gradient(pol, traj)[0]
    assert rel_err(g_pol, policy_grad_f64(pol.get_params(), want)) < GRAD_RTOL
    p = cri.get_params()
    g_cri, loss_d = ra.critic_gradient(cri, traj)
    v, _ = O.gru_seq_forward(CS, p, want, f64=True, want_succ=False)
    d = v - want["rtg"].astype(np.float64)[None]
    assert rel_err(g_cri, O.gru_seq_backward(CS, p, want, 2.0 * d / v[0].size, f64=True)) < GRAD_RTOL
    assert abs(loss_d - (d * d).mean()) <= 1e-5 * max((d * d).mean(), 1e-12)


def test_bf16_pipe_training_passes_agree_with_the_f32_kernels(engine):
    """the two builds of the GRU chain's training passes — recurrence on the bf16 matrix pipe with exact three-piece
    products (kernels_seq_train.hip, the default) and round 1's f32 kernels (engine kernel variant 1) — give the same
    gradients and Fisher-vector products up to the order of their f32 sums"""
    pol, cri, traj, want, _, _ = setup_update(engine)
    v = np.random.default_rng(9).normal(size=pol.P).astype(np.float32)
    got = {}
    for variant in (0, 1):
        engine.set_kernel_variant(variant)
        try:
            got[variant] = (ra.policy_gradient(pol, traj)[0], ra.critic_gradient(cri, traj)[0],
                            ra.policy_fvp(pol, traj, v, 0.0))
        finally:
            engine.set_kernel_variant(0)
    for a, b in zip(got[0], got[1]):
        assert np.abs(a).max() > 0 and rel_err(a, b.astype(np.float64)) < GRAD_RTOL


def test_piece_products_keep_f32_accuracy(engine):
    """the bf16 pipe multiplies the piece pairs (p, q) with p + q < 3 of the three-piece splits — six of nine; the
    dropped pairs are below 2^-26 |a||b| per term, a quarter of the rounding of one f32 product — so its gradients
    must sit as close to the f64 oracle's as those of the f32 kernels (sequential fma chains) do"""
    pol, cri, traj, want, _, _ = setup_update(engine)
    g64 = {"policy": policy_grad_f64(pol.get_params(), want)}
    v, _ = O.gru_seq_forward(CS, cri.get_params(), want, f64=True, want_succ=False)
    d = v - want["rtg"].astype(np.float64)[None]
    g64["critic"] = O.gru_seq_backward(CS, cri.get_params(), want, 2.0 * d / v[0].size, f64=True)
    err = {}
    for variant in (0, 1):
        engine.set_kernel_variant(variant)
        try:
            err[variant] = {"policy": rel_err(ra.policy_gradient(pol, traj)[0], g64["policy"]),
                            "critic": rel_err(ra.critic_gradient(cri, traj)[0], g64["critic"])}
        finally:
            engine.set_kernel_variant(0)
    print("relative error against f64 (bf16 pieces, f32 kernels):", err)
    for k in ("policy", "critic"):
        assert err[0][k] < GRAD_RTOL
        assert err[0][k] < 2.0 * err[1][k] + 1e-7, err


def oracle_ppo_loop(p, want, ocfg, steps, clip):
    p = p.copy()
    st = L.oracle_adam_new(len(p))
    logits, _ = O.gru_seq_forward(PS, p, want, want_succ=False)
    _, lp0, _, ent_sum = O.seq_policy_dlogits(logits, want["action"], want["adv"])
    B = want["action"].size
    losses = []
    for k in range(steps):
        logits, _ = O.gru_seq_forward(PS, p, want, want_succ=False)
        dl, _, obj, _ = O.seq_policy_dlogits(logits, want["action"], want["adv"], logp0=lp0, clip=clip)
        losses.append(-obj / B)
        g = O.gru_seq_backward(PS, p, want, dl)
        L.oracle_adam_step_f32(st, C.byref(ocfg), O.f32p(p), O.f32p(g))
    L.oracle_adam_free(st)
    return p, np.array(losses), ent_sum / B


@pytest.mark.parametrize("lr,steps", [(1e-3, 4), (1e-2, 5)], ids=["default-lr", "clipping-active"])
def test_ppo_update_recurrent(engine, lr, steps):
    pol, cri, traj, want, acfg, ocfg = setup_update(engine, lr=lr)
    p0 = pol.get_params().copy()
    opt = ra.Adam(pol, acfg)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = steps
    st, losses_d = ra.ppo_update(pol, opt, traj, cfg, want_losses=True)
    p_o, losses_o, ent_o = oracle_ppo_loop(p0, want, ocfg, steps, 0.2)
    assert abs(st.entropy - ent_o) < 1e-5
    assert np.max(np.abs(losses_d - losses_o)) < 3e-5 * max(1.0, np.abs(losses_o).max()), (losses_d, losses_o)
    diff = np.abs(pol.get_params() - p_o)
    assert np.mean(diff < 3e-5) > 0.97  # Adam amplifies rounding-level gradient differences where |g| ~ 0
    assert losses_d[-1] < losses_d[0]


def test_reinforce_and_critic_updates_recurrent(engine):
    pol, cri, traj, want, acfg, ocfg = setup_update(engine)
    # REINFORCE: one Adam step along the surrogate gradient
    p0 = pol.get_params().copy()
    st = ra.reinforce_update(pol, ra.Adam(pol, acfg), traj)
    logits, _ = O.gru_seq_forward(PS, p0, want, want_succ=False)
    dl, lp, _, ent_sum = O.seq_policy_dlogits(logits, want["action"], want["adv"])
    B = want["action"].size
    assert abs(st.loss_first + float((lp.astype(np.float64) * want["adv"]).sum()) / B) < 1e-5
    assert abs(st.entropy - ent_sum / B) < 1e-5
    p_o = p0.copy()
    ost = L.oracle_adam_new(len(p_o))
    L.oracle_adam_step_f32(ost, C.byref(ocfg), O.f32p(p_o), O.f32p(O.gru_seq_backward(PS, p0, want, dl)))
    L.oracle_adam_free(ost)
    moved = np.abs(pol.get_params() - p0)
    assert 0 < moved.max() <= 1.001e-3  # first Adam step: at most lr per parameter
    assert np.mean(np.abs(pol.get_params() - p_o) < 3e-5) > 0.97
    # critic: 3 x {MSE, backward through time, Adam}
    c0 = cri.get_params().copy()
    cst, losses_d = ra.critic_update(cri, ra.Adam(cri, acfg), traj, 3, want_losses=True)
    p = c0.copy()
    ost = L.oracle_adam_new(len(p))
    losses_o = []
    for k in range(3):
        v, _ = O.gru_seq_forward(CS, p, want, want_succ=False)
        d = v[0] - want["rtg"]
        losses_o.append(float((d.astype(np.float64) ** 2).mean()))
        g = O.gru_seq_backward(CS, p, want, (d * np.float32(2.0 / B))[None])
        L.oracle_adam_step_f32(ost, C.byref(ocfg), O.f32p(p), O.f32p(g))
    L.oracle_adam_free(ost)
    assert np.max(np.abs(losses_d - np.array(losses_o)) / np.array(losses_o)) < 2e-5
    assert np.mean(np.abs(cri.get_params() - p) < 3e-5) > 0.97
    assert losses_d[-1] < losses_d[0]


def test_fisher_vector_product_through_time(engine):
    pol, cri, traj, want, _, _ = setup_update(engine)
    p = pol.get_params()
    rng = np.random.default_rng(3)
    v = rng.normal(size=len(p)).astype(np.float32)
    reg = 1e-5
    hv_d = ra.policy_fvp(pol, traj, v, reg)
    hv64 = O.gru_policy_fvp(PS, p.astype(np.float64), v.astype(np.float64), want, reg, f64=True)
    hv32 = O.gru_policy_fvp(PS, p, v, want, reg)
    assert rel_err(hv_d, hv64) < 2e-5, (rel_err(hv_d, hv64), rel_err(hv32, hv64))
    assert rel_err(hv_d, hv64) < 4 * rel_err(hv32, hv64) + 2e-6
    # loss / KL of perturbed parameters against the old policy
    p1 = p + 0.05 * v / np.abs(v).max()
    pol.set_params(p1)
    loss_d, kl_d = ra.policy_loss_kl(pol, traj, p)
    l0, _ = O.gru_seq_forward(PS, p, want, f64=True, want_succ=False)
    l1, _ = O.gru_seq_forward(PS, p1, want, f64=True, want_succ=False)
    lp0, lp1 = l0 - np.log(np.exp(l0).sum(0)), l1 - np.log(np.exp(l1).sum(0))
    a = want["action"].astype(np.int64)
    sel = lambda lp: np.where(a == 0, lp[0], lp[1])
    loss_o = -(np.exp(sel(lp1) - sel(lp0)) * want["adv"]).mean()
    kl_o = (np.exp(lp0) * (lp0 - lp1)).sum(0).mean()
    assert abs(loss_d - loss_o) <= 2e-5 * max(1.0, abs(loss_o))
    # per-sample KL terms are f32 differences of log-probabilities: absolute error ~1e-8 on the mean
    assert abs(kl_d - kl_o) <= 1e-4 * kl_o + 3e-8 and kl_o > 1e-5


def oracle_trpo_loop(p0, want, f64, iterations=10, max_backtracks=15, ratio=0.8, reg=1e-5, max_kl=0.01):
    """trust_region_backward_step + backtracking_line_search (conjugate_gradient.rs:115-255) on the oracle's
    recurrent gradient / Fisher-vector / loss-KL pieces; arithmetic in f32 or f64"""
    dt = np.float64 if f64 else np.float32
    p0 = p0.astype(dt)
    a = want["action"].astype(np.int64)
    adv = want["adv"].astype(dt)
    B = a.size

    def logp(p):
        l, _ = O.gru_seq_forward(PS, p, want, f64=f64, want_succ=False)
        l = l.astype(dt)
        z = l - l.max(0)
        return (z - np.log(np.exp(z).sum(0))).astype(dt)

    lp0 = logp(p0)
    sel = lambda lp: np.where(a == 0, lp[0], lp[1])
    pr = np.exp(lp0)
    ind = np.stack([a == 0, a == 1]).astype(dt)
    g = O.gru_seq_backward(PS, p0, want, (-(adv / dt(B)) * (ind - pr)).astype(dt), f64=f64).astype(dt)
    loss0 = dt(-(adv.astype(np.float64)).mean())
    fvp = lambda v: O.gru_policy_fvp(PS, p0, v, want, reg, f64=f64).astype(dt)
    x = np.zeros_like(g)
    r, pp = g.copy(), g.copy()
    rr = dt(r @ r)
    iters = 0
    for _ in range(iterations):
        z = fvp(pp)
        alpha = dt(rr / dt(pp @ z))
        x = (x + alpha * pp).astype(dt)
        r = (r - alpha * z).astype(dt)
        new_rr = dt(r @ r)
        iters += 1
        if float(new_rr) < 1e-10:
            break
        pp = (r + dt(new_rr / rr) * pp).astype(dt)
        rr = new_rr
    xhx = float(x @ fvp(x))
    step_size = np.sqrt(1.0 / (xhx + 1e-8) * max_kl * 2.0)
    descent = (dt(step_size) * x).astype(dt)
    for i in range(max_backtracks):
        cand = (p0 - dt(ratio ** i) * descent).astype(dt)
        lp1 = logp(cand)
        loss = -(np.exp(sel(lp1) - sel(lp0)).astype(np.float64) * adv).mean()
        kl = (np.exp(lp0).astype(np.float64) * (lp0 - lp1)).sum(0).mean()
        if loss < float(loss0) and kl <= max_kl:
            return cand, step_size, i, loss, kl, iters
    return p0, step_size, -1, float(loss0), 0.0, iters


def test_trpo_update_through_time(engine):
    """Same tolerance rule as the feed-forward TRPO test: f32 CG is ill-conditioned, so the device result is
    compared with the f64 evaluation and must be no farther from it than twice the f32 restatement is."""
    pol, cri, traj, want, _, _ = setup_update(engine)
    p0 = pol.get_params().copy()
    st = ra.trpo_update(pol, traj)
    p64, ss64, bt64, loss64, kl64, it64 = oracle_trpo_loop(p0, want, True)
    p32, ss32, bt32, loss32, kl32, it32 = oracle_trpo_loop(p0, want, False)
    assert st.status == ra.OPT_OK and st.cg_iterations == it64 == it32 == 10
    assert abs(st.step_size - ss64) <= 2.0 * abs(ss32 - ss64) + 1e-3 * ss64, (st.step_size, ss32, ss64)
    assert min(bt32, bt64) - 1 <= st.num_backtracks <= max(bt32, bt64) + 1
    assert st.constraint_val_final <= 0.01 and st.loss_final < st.loss_initial
    # what was accepted is what the device says: re-evaluate the new parameters independently
    p1 = pol.get_params()
    loss_d, kl_d = ra.policy_loss_kl(pol, traj, p0)
    assert abs(loss_d - st.loss_final) <= 1e-5 * max(1.0, abs(st.loss_final))
    assert abs(kl_d - st.constraint_val_final) <= 1e-5 + 1e-3 * kl_d
    assert np.abs(p1 - p0).max() > 0


# ------------------------------------------------------------------ every env kind with every module kind
def test_cartpole_lanes_with_the_recurrent_policy(engine):
    n, T = 64, 40
    env = ra.CartPoleEnv(engine, n, max_steps=25, seed_env=5, seed_actor=6)
    sim = O.LaneSim(n, max_steps=25, seed_env=5, seed_actor=6)
    pol = ra.GruMlp(engine, 5, 2)
    pol.init(31)
    traj = ra.Trajectory(engine, n, T, 5)
    for period in range(2):
        ra.rollout(env, pol, traj)
        want = sim.rollout_gru(PS, pol.get_params(), T)
        got = traj.read_all()
        for k in ("obs", "action", "reward", "flag"):
            assert np.array_equal(got[k], want[k]), (period, k)
    assert (want["flag"] == O.TERMINATE).any() and (want["flag"] == O.INTERRUPT).any()
    for a, b in zip(env.get_state(), sim.get_state()):
        assert np.array_equal(a, b)
    lo_d, _ = pol.seq_forward(traj, want_succ=False)
    lo_o, _ = O.gru_seq_forward(PS, pol.get_params(), want, want_succ=False)
    assert np.array_equal(lo_d, lo_o)
    ra.reward_to_go(traj, 0.99)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 3
    st, losses = ra.ppo_update(pol, ra.Adam(pol), traj, cfg, want_losses=True)
    assert losses[-1] < losses[0]


def test_chain_lanes_with_the_feed_forward_policy(engine):
    n, T = 200, 45
    ms, cs = O.MlpShape(5, 128, 2), O.MlpShape(5, 128, 1)
    env, sim = chain_pair(engine, n, max_steps=9)
    pol, cri = ra.Mlp(engine, 5, 128, 2), ra.Mlp(engine, 5, 128, 1)
    pol.init(41)
    cri.init(42)
    traj = ra.Trajectory(engine, n, T, 5)
    for period in range(2):
        ra.rollout(env, pol, traj)
        want = sim.rollout_mlp(ms, pol.get_params(), T)
        got = traj.read_all()
        for k in ("obs", "action", "reward", "flag"):
            assert np.array_equal(got[k], want[k]), (period, k)
        m = want["flag"] == O.INTERRUPT
        assert m.any() and np.array_equal(got["term_obs"][:, m], want["term_obs"][:, m])
    ra.gae(traj, cri, 0.95, 0.9)
    v_o, adv_o, rtg_o = O.lanes_gae(cs, cri.get_params(), want, np.float32(0.95), np.float32(0.9))
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), adv_o) and np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)
    st = ra.trpo_update(pol, traj)
    assert st.status == ra.OPT_OK and st.constraint_val_final <= 0.01 and st.loss_final < st.loss_initial


# ------------------------------------------------------------------ chains of other widths (RnnBaseConfig::hidden_size,
# ChainConfig::hidden_dim, MlpConfig::hidden_sizes = [h]; src/torch/modules/seq/rnn/mod.rs:20-45, modules/chain.rs:19-32)
# run on the 5 -> 128 -> 128 kernels embedded with zero padding (rl_mlp::exec, relearn_amd/csrc/engine.hpp)
NARROW = [("gru", 3, 4, 8, 2),      # the GRU(3 -> 4) of the reference's benches/rnn.rs:14-65, with a head
          ("gru", 5, 32, 64, 1), ("lstm", 4, 20, 16, 2), ("gru", 5, 128, 40, 2), ("lstm", 1, 1, 1, 1),
          ("gru", 2, 127, 128, 2)]


def narrow_modules(engine, cell, D, H, H2, A, seed):
    cls = ra.GruMlp if cell == "gru" else ra.LstmMlp
    m = cls(engine, D, A, H, H2)
    m.init(seed)
    shape = O.GruShape(D, H, H2, A, O.CELL_GRU if cell == "gru" else O.CELL_LSTM)
    assert m.P == L.oracle_gru_num_params(shape)
    assert np.array_equal(m.get_params(), O.gru_init(shape, seed))  # RnnWeights::new on the narrow shape, bit for bit
    return m, shape


def synthetic_history(engine, n, T, D, seed):
    """a host-made lane history of obs_dim D (no env of this path has fewer than four features): observations,
    successor codes with all three kinds, successor observations, actions, advantages, returns"""
    rng = np.random.default_rng(seed)
    want = {"obs": rng.normal(size=(D, T + 1, n)).astype(np.float32),
            "flag": rng.choice(np.array([0, 0, 0, 0, 1, 2], dtype=np.uint8), size=(T, n)),
            "term_obs": rng.normal(size=(D, T, n)).astype(np.float32),
            "action": rng.integers(0, 2, size=(T, n)).astype(np.uint8),
            "reward": rng.normal(size=(T, n)).astype(np.float32)}
    traj = ra.Trajectory(engine, n, T, D)
    traj.write_all(want)
    want["adv"] = rng.normal(size=(T, n)).astype(np.float32)
    want["rtg"] = rng.normal(size=(T, n)).astype(np.float32)
    traj.write(ra.TRAJ_ADVANTAGES, want["adv"])
    traj.write(ra.TRAJ_RETURNS, want["rtg"])
    return traj, want


@pytest.mark.parametrize("cell,D,H,H2,A", NARROW)
def test_narrow_chains_forward_is_bit_exact(engine, cell, D, H, H2, A):
    """module outputs and successor outputs of a narrow chain on the padded kernels against the oracle run on the
    NARROW shape: identical bits (padding adds terms fma(0, 0, acc) to each dot product and nothing else)"""
    m, shape = narrow_modules(engine, cell, D, H, H2, A, 31)
    traj, want = synthetic_history(engine, 64, 12, D, 5)
    out_d, succ_d = m.seq_forward(traj)
    out_o, succ_o = O.gru_seq_forward(shape, m.get_params(), want)
    assert np.array_equal(out_d, out_o) and np.array_equal(succ_d, succ_o)
    assert np.abs(out_o).max() > 0 and np.count_nonzero(succ_o) > 0
    # the parameters went through the twin and came back untouched
    assert np.array_equal(m.get_params(), O.gru_init(shape, 31))
    if A == 1:  # values, advantages and returns through the recurrent critic path
        ra.gae(traj, m, 0.95, 0.9)
        adv_o, rtg_o = O.seq_gae(out_o[0], succ_o[0], want, np.float32(0.95), np.float32(0.9))
        assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), adv_o) and np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)


@pytest.mark.parametrize("cell,D,H,H2,A", NARROW)
def test_narrow_chains_gradients_against_f64(engine, cell, D, H, H2, A):
    """surrogate / critic gradient through time and (policies) a Fisher-vector product of the narrow chain, gathered from
    the padded kernels' layout into the module's flat order, against the f64 oracle of the narrow shape"""
    m, shape = narrow_modules(engine, cell, D, H, H2, A, 41)
    traj, want = synthetic_history(engine, 96, 9, D, 6)
    p = m.get_params()
    B = want["action"].size
    if A == 2:
        g_d, loss_d, ent_d = ra.policy_gradient(m, traj)
        logits, _ = O.gru_seq_forward(shape, p, want, f64=True, want_succ=False)
        z = logits - logits.max(0)
        lp = z - np.log(np.exp(z).sum(0))
        pr = np.exp(lp)
        a = want["action"].astype(np.int64)
        ind = np.stack([a == 0, a == 1]).astype(np.float64)
        g64 = O.gru_seq_backward(shape, p, want, -(want["adv"].astype(np.float64) / B) * (ind - pr), f64=True)
        assert g_d.shape == g64.shape and rel_err(g_d, g64) < GRAD_RTOL
        v = np.random.default_rng(3).normal(size=m.P).astype(np.float32)
        h_d = ra.policy_fvp(m, traj, v, 1e-5)
        h64 = O.gru_policy_fvp(shape, p, v, want, 1e-5, f64=True)
        assert rel_err(h_d, h64) < 2e-5
    else:
        g_d, loss_d = ra.critic_gradient(m, traj)
        v, _ = O.gru_seq_forward(shape, p, want, f64=True, want_succ=False)
        d = v - want["rtg"].astype(np.float64)[None]
        g64 = O.gru_seq_backward(shape, p, want, 2.0 * d / B, f64=True)
        assert rel_err(g_d, g64) < GRAD_RTOL
        assert abs(loss_d - (d * d).mean()) <= 1e-5 * (d * d).mean()
        # and a few Adam steps move the narrow parameter vector (the optimiser works on the flat order)
        st, losses = ra.critic_update(m, ra.Adam(m), traj, 3, want_losses=True)
        assert losses[-1] < losses[0] and not np.array_equal(m.get_params(), p)


def test_narrow_chain_rolls_out_and_learns_on_chain_lanes(engine):
    """in_dim 5 with narrow widths runs the fused recurrent rollout: bit-exact against the oracle's lanes with the narrow
    module, and PPO steps through the padded kernels lower its loss"""
    n, T = 64, 30
    env, sim = chain_pair(engine, n, max_steps=9)
    pol, shape = narrow_modules(engine, "gru", 5, 24, 48, 2, 11)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    want = sim.rollout_gru(shape, pol.get_params(), T)
    got = traj.read_all()
    for k in ("obs", "action", "reward", "flag"):
        assert np.array_equal(got[k], want[k]), k
    ra.reward_to_go(traj, 0.95)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 4
    st, losses = ra.ppo_update(pol, ra.Adam(pol), traj, cfg, want_losses=True)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_recurrent_shapes_that_are_not_built_are_refused(engine):
    h = C.c_void_p()
    assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(0), C.c_uint32(5), C.c_uint32(32), C.c_uint32(5),
                                      C.c_uint32(32), C.c_uint32(2), C.byref(h)) == ra.ERR_UNSUPPORTED  # num_layers = 5
    assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(0), C.c_uint32(5), C.c_uint32(32), C.c_uint32(0),
                                      C.c_uint32(32), C.c_uint32(2), C.byref(h)) == ra.ERR_BUILD_AGENT
    for bad in ((9, 32, 32, 2), (5, 257, 32, 2), (5, 32, 300, 2), (5, 32, 32, 3), (0, 4, 4, 1)):
        assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(1), C.c_uint32(bad[0]), C.c_uint32(bad[1]), C.c_uint32(1),
                                          C.c_uint32(bad[2]), C.c_uint32(bad[3]), C.byref(h)) == ra.ERR_BUILD_AGENT
    assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(1), C.c_uint32(4), C.c_uint32(7), C.c_uint32(1), C.c_uint32(9),
                                      C.c_uint32(1), C.byref(h)) == ra.OK and h
    ra.lib().rl_mlp_destroy(h)
