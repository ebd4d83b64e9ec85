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
    with pytest.raises(ra.RelearnError) as e:  # (other widths are built since round 4: tests/test_gpu_gru.py, test_gpu_stacked.py)
        ra.LstmMlp(engine, 5, 2, lstm_hidden=257)
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


# ------------------------------------------------------------------ gradients through time and the update loops
GRAD_RTOL = 5e-6   # device f32 (MFMA partial sums over <= 2048 samples, then f64) vs the f64 evaluation, rel. to max|g|


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def setup_update(engine, n=64, T=30, max_steps=9, lr=1e-3):
    env = ra.ChainEnv(engine, n, max_steps=max_steps, seed_env=3, seed_actor=4)
    sim = O.ChainLaneSim(n, max_steps=max_steps, seed_env=3, seed_actor=4)
    pol, cri = ra.LstmMlp(engine, 5, 2), ra.LstmMlp(engine, 5, 1)
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


def test_policy_gradient_through_time(engine):
    pol, cri, traj, want, _, _ = setup_update(engine)
    p = pol.get_params()
    g_d, loss_d, ent_d = ra.policy_gradient(pol, traj)
    # -mean(ratio * A) at ratio 1 through time, everything in f64 (numpy softmax + the f64 BPTT oracle)
    logits, _ = O.gru_seq_forward(PS, p, want, f64=True, want_succ=False)
    B = logits[0].size
    z = logits - logits.max(0)
    lp = z - np.log(np.exp(z).sum(0))
    a = want["action"].astype(np.int64)
    ind = np.stack([a == 0, a == 1]).astype(np.float64)
    g64 = O.gru_seq_backward(PS, p, want, -(want["adv"].astype(np.float64) / B) * (ind - np.exp(lp)), f64=True)
    logits32, _ = O.gru_seq_forward(PS, p, want, want_succ=False)
    _, _, loss_sum, ent_sum = O.seq_policy_dlogits(logits32, want["action"], want["adv"])
    assert rel_err(g_d, g64) < GRAD_RTOL, rel_err(g_d, g64)
    assert abs(loss_d + loss_sum / B) <= 1e-5 * max(1.0, abs(loss_sum / B))
    assert abs(ent_d - ent_sum / B) < 1e-5
    H, D = 128, 5  # every parameter block receives gradient: W_ih, W_hh, b_ih, b_hh (4 gate blocks each), W1, b1, W2, b2
    cuts = np.cumsum([4 * H * D, 4 * H * H, 4 * H, 4 * H, H * H, H, 2 * H, 2])
    for lo, hi in zip(np.r_[0, cuts[:-1]], cuts):
        assert np.abs(g_d[lo:hi]).max() > 0
        assert rel_err(g_d[lo:hi], g64[lo:hi]) < 50 * GRAD_RTOL
    for gate in range(4):  # and every gate of the recurrent matrix
        blk = slice(4 * H * D + gate * H * H, 4 * H * D + (gate + 1) * H * H)
        assert np.abs(g_d[blk]).max() > 0 and rel_err(g_d[blk], g64[blk]) < 50 * GRAD_RTOL


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


@pytest.mark.parametrize("n,T,max_steps", [(32, 1, 9), (32, 2, 1), (96, 7, 3), (2048, 12, 4)],
                         ids=["one-tile-one-step", "every-step-ends", "three-tiles", "two-blocks-per-chunk"])
def test_critic_gradient_at_edge_shapes(engine, n, T, max_steps):
    """the LSTM chain's forward record, BPTT and weight-gradient kernels at the edges of their loops"""
    pol, cri, traj, want, _, _ = setup_update(engine, n=n, T=T, max_steps=max_steps)
    p = cri.get_params()
    g_d, loss_d = ra.critic_gradient(cri, traj)
    v, _ = O.gru_seq_forward(CS, p, want, f64=True, want_succ=False)
    d = v - want["rtg"].astype(np.float64)[None]
    assert rel_err(g_d, O.gru_seq_backward(CS, p, want, 2.0 * d / v[0].size, f64=True)) < GRAD_RTOL
    assert abs(loss_d - (d * d).mean()) <= 1e-5 * max((d * d).mean(), 1e-12)


def test_ppo_and_critic_updates(engine):
    pol, cri, traj, want, acfg, ocfg = setup_update(engine)
    p0 = pol.get_params().copy()
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 4
    st, losses_d = ra.ppo_update(pol, ra.Adam(pol, acfg), traj, cfg, want_losses=True)
    # the same loop on the oracle: clipped surrogate against log pi_0, backward through time, Adam
    p = p0.copy()
    ost = L.oracle_adam_new(len(p))
    logits, _ = O.gru_seq_forward(PS, p, want, want_succ=False)
    _, lp0, _, ent_sum = O.seq_policy_dlogits(logits, want["action"], want["adv"])
    B = want["action"].size
    losses_o = []
    for k in range(4):
        logits, _ = O.gru_seq_forward(PS, p, want, want_succ=False)
        dl, _, obj, _ = O.seq_policy_dlogits(logits, want["action"], want["adv"], logp0=lp0, clip=0.2)
        losses_o.append(-obj / B)
        L.oracle_adam_step_f32(ost, C.byref(ocfg), O.f32p(p), O.f32p(O.gru_seq_backward(PS, p, want, dl)))
    L.oracle_adam_free(ost)
    assert abs(st.entropy - ent_sum / B) < 1e-5
    assert np.max(np.abs(losses_d - np.array(losses_o))) < 3e-5 * max(1.0, np.abs(losses_o).max())
    assert np.mean(np.abs(pol.get_params() - p) < 3e-5) > 0.97  # Adam amplifies rounding-level differences where |g| ~ 0
    assert losses_d[-1] < losses_d[0]
    # critic: 3 x {MSE, backward through time, Adam}
    c0 = cri.get_params().copy()
    cst, closs_d = ra.critic_update(cri, ra.Adam(cri, acfg), traj, 3, want_losses=True)
    p = c0.copy()
    ost = L.oracle_adam_new(len(p))
    closs_o = []
    for k in range(3):
        v, _ = O.gru_seq_forward(CS, p, want, want_succ=False)
        d = v[0] - want["rtg"]
        closs_o.append(float((d.astype(np.float64) ** 2).mean()))
        L.oracle_adam_step_f32(ost, C.byref(ocfg), O.f32p(p), O.f32p(O.gru_seq_backward(CS, p, want, (d * np.float32(2.0 / B))[None])))
    L.oracle_adam_free(ost)
    assert np.max(np.abs(closs_d - np.array(closs_o)) / np.array(closs_o)) < 2e-5
    assert np.mean(np.abs(cri.get_params() - p) < 3e-5) > 0.97 and closs_d[-1] < closs_d[0]


def test_bf16_pipe_training_passes_agree_with_the_f32_kernels(engine):
    """the two builds of the LSTM chain's training passes — forward recurrence, head and weight gradients on the bf16 matrix
    pipe with exact three-piece products (kernels_seq_train.hip: k_lstm_recur_fwd on four waves per tile, the GRU chain's
    head and weight-gradient kernels with four gate blocks; the default) and round 1's f32 kernels (engine kernel variant
    1) — give the same gradients and Fisher-vector products up to the order of their f32 sums and the 2-ulp gate
    functions of the training forward"""
    pol, cri, traj, want, _, _ = setup_update(engine, n=96, T=24, max_steps=7)
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
        assert np.abs(a).max() > 0 and rel_err(a, b.astype(np.float64)) < 5e-6


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


def test_trpo_update_through_time(engine):
    """conjugate-gradient trust-region step through the LSTM: accepted inside the region, and what the device reports
    is what an independent re-evaluation of the new parameters gives (the CG arithmetic itself is compared with the
    oracle's in tests/test_gpu_gru.py: the same device code runs here on the LSTM's gradient / Fisher-vector passes)"""
    pol, cri, traj, want, _, _ = setup_update(engine)
    p0 = pol.get_params().copy()
    st = ra.trpo_update(pol, traj)
    assert st.status == ra.OPT_OK and st.cg_iterations == 10
    assert st.constraint_val_final <= 0.01 and st.loss_final < st.loss_initial
    loss_d, kl_d = ra.policy_loss_kl(pol, traj, p0)
    assert abs(loss_d - st.loss_final) <= 1e-5 * max(1.0, abs(st.loss_final))
    assert abs(kl_d - st.constraint_val_final) <= 1e-5 + 1e-3 * kl_d
    # against the f64 evaluation of the same accepted step
    l0, _ = O.gru_seq_forward(PS, p0, want, f64=True, want_succ=False)
    l1, _ = O.gru_seq_forward(PS, pol.get_params(), want, f64=True, want_succ=False)
    lp0, lp1 = l0 - np.log(np.exp(l0).sum(0)), l1 - np.log(np.exp(l1).sum(0))
    a = want["action"].astype(np.int64)
    sel = lambda lp: np.where(a == 0, lp[0], lp[1])
    assert abs(-(np.exp(sel(lp1) - sel(lp0)) * want["adv"]).mean() - st.loss_final) <= 2e-5 * max(1.0, abs(st.loss_final))
    assert abs((np.exp(lp0) * (lp0 - lp1)).sum(0).mean() - st.constraint_val_final) <= 1e-4 * st.constraint_val_final + 3e-8


def test_the_lstm_policy_learns_the_memory_game(engine):
    """the behavioural check of tests/test_gpu_memory.py with the LSTM cell: answering needs the first observation"""
    n, T = 2048, 32
    env = ra.MemoryEnv(engine, n, seed_env=21, seed_actor=22)
    pol = ra.LstmMlp(engine, 5, 2)
    pol.init(5)
    traj = ra.Trajectory(engine, n, T, 5)
    acfg = ra.adam_config_default()
    acfg.learning_rate = 3e-3
    opt = ra.Adam(pol, acfg)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 5
    acc = []
    for _ in range(26):
        ra.rollout(env, pol, traj)
        r = traj.read(ra.TRAJ_REWARD)
        acc.append(float((r == 1.0).sum()) / float((r != 0.0).sum()))
        ra.reward_to_go(traj, 1.0)
        ra.ppo_update(pol, opt, traj, cfg)
    assert 0.4 < acc[0] < 0.6 and acc[-1] >= 0.9, acc


def test_successor_evaluations_leave_the_state_of_the_other_lanes_alone(engine):
    """A history whose lanes are interrupted at DIFFERENT steps (what CartPole lanes under a step limit produce once
    episodes have ended at different times): the successor evaluation of an interrupted lane runs as an extra iteration
    of its whole tile, and every other lane must enter the next step with the state it had.  (Round 4: the LSTM cell
    parked the head's activations in the state buffer that iteration had just read; the env-made histories of the other
    tests interrupt all lanes of a tile at once, where the state is reset anyway.)  Outputs and successor outputs bit
    for bit against the oracle, for both cells."""
    from test_gpu_gru import synthetic_history
    for cls, shape in ((ra.LstmMlp, O.LstmShape(5, 128, 128, 1)), (ra.GruMlp, O.GruShape(5, 128, 128, 2))):
        m = cls(engine, 5, shape.out_dim)
        m.init(17)
        traj, want = synthetic_history(engine, 96, 20, 5, 8)
        out_d, succ_d = m.seq_forward(traj)
        out_o, succ_o = O.gru_seq_forward(shape, m.get_params(), want)
        assert np.array_equal(out_d, out_o) and np.array_equal(succ_d, succ_o)
        assert 0 < (want["flag"] == O.INTERRUPT).mean() < 0.5
