"""Parity at BASELINE.json's full sizes, through properties that do not need a full-size oracle run:
lanes are independent, so slices of a full-size device rollout are compared bit for bit with the oracle run on just
those lanes (same global lane ids); sharding invariance; linearity of sample sums; the trust-region acceptance rule;
replay-ring invariants; agreement of the parallel and the sequential minibatch samplers."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

H = 128
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)
N, T = 65536, 128  # configs[1] / configs[3]: 65,536 CartPole lanes, horizon 128


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(scope="module")
def full(engine):
    env = ra.CartPoleEnv(engine, N, max_steps=500, seed_env=0, seed_actor=1)
    pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
    pol.init(2)
    cri.init(3)
    traj = ra.Trajectory(engine, N, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    return dict(env=env, pol=pol, cri=cri, traj=traj, data=traj.read_all(), adv=traj.read(ra.TRAJ_ADVANTAGES),
                rtg=traj.read(ra.TRAJ_RETURNS), values=traj.read(ra.TRAJ_VALUES))


@pytest.mark.parametrize("lo", [0, 31337, N - 96])
def test_full_size_rollout_slices_equal_the_oracle(full, lo):
    n = 96
    sim = O.LaneSim(n, max_steps=500, lane_offset=lo, seed_env=0, seed_actor=1)
    want = sim.rollout(PS, full["pol"].get_params(), T)
    got = full["data"]
    for k in ("obs", "action", "reward", "flag"):
        assert np.array_equal(got[k][..., lo:lo + n], want[k]), k
    v_o, adv_o, rtg_o = O.lanes_gae(CS, full["cri"].get_params(), want, np.float32(0.99), np.float32(0.95))
    assert np.array_equal(full["values"][:, lo:lo + n], v_o)
    assert np.array_equal(full["adv"][:, lo:lo + n], adv_o)
    assert np.array_equal(full["rtg"][:, lo:lo + n], rtg_o)


def test_full_size_sharding_invariance_and_gradient_linearity(engine, full):
    """two 32,768-lane shards (what two GPUs hold) reproduce the full rollout bit for bit; the full-batch gradient is
    the sample-weighted mean of the shard gradients"""
    halves = []
    for r in range(2):
        env = ra.CartPoleEnv(engine, N // 2, max_steps=500, lane_offset=r * (N // 2), seed_env=0, seed_actor=1)
        traj = ra.Trajectory(engine, N // 2, T, 5)
        ra.rollout(env, full["pol"], traj)
        ra.gae(traj, full["cri"], 0.99, 0.95)
        assert np.array_equal(traj.read(ra.TRAJ_ACTION), full["data"]["action"][:, r * (N // 2):(r + 1) * (N // 2)])
        assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), full["adv"][:, r * (N // 2):(r + 1) * (N // 2)])
        g, loss, ent = ra.policy_gradient(full["pol"], traj)
        gc, lc = ra.critic_gradient(full["cri"], traj)
        halves.append((g, loss, ent, gc, lc))
    g, loss, ent = ra.policy_gradient(full["pol"], full["traj"])
    gc, lc = ra.critic_gradient(full["cri"], full["traj"])
    assert rel_err(0.5 * (halves[0][0] + halves[1][0]), g) < 2e-6
    assert rel_err(0.5 * (halves[0][3] + halves[1][3]), gc) < 2e-6
    assert abs(0.5 * (halves[0][1] + halves[1][1]) - loss) < 1e-6 * max(1.0, abs(loss))
    assert abs(0.5 * (halves[0][4] + halves[1][4]) - lc) < 1e-5 * lc
    assert abs(0.5 * (halves[0][2] + halves[1][2]) - ent) < 1e-6


def test_full_size_update_kernels_against_the_f64_oracle(engine, full):
    """The update kernels at the size the bench runs them — 65,536 x 128 = 8.39 M samples: 256 persistent workgroups,
    1,024 tiles per wave, 16 f32 -> f64 flushes per wave — against the oracle's f64 ground truth over ALL samples
    (oracle_grad_f64_mt: the f64 functions of oracle/nn_impl.inc, OpenMP over chunks): one Fisher-vector product and
    the critic gradient within 1e-6 of the vector's largest entry (measured: 5e-8, 4e-8); the surrogate gradient by the
    criterion of tests/test_gpu_parity.py::test_policy_gradient_matches_oracle — no farther from the f64 truth than
    4 x what the f32 ORACLE is, evaluated over the same 8.39 M samples (oracle_grad_f32_mt).  That gradient is a sum
    of terms of either sign, -A (1[a] - p) / B with mean(A) = 8, that cancel to ~ B^-1/2 of their size; per-sample f32
    arithmetic leaves 1.4e-6 of max|g| in it (the f32 oracle), the device 2.8e-6.
    Reference: trpo.rs:124-131, conjugate_gradient.rs:312-338, critics/opt.rs:109-115."""
    x, a = O.flat_samples(full["data"])
    adv, rtg = full["adv"].reshape(-1), full["rtg"].reshape(-1)
    pp, cp = full["pol"].get_params(), full["cri"].get_params()
    assert len(a) == N * T
    g_d, loss_d, ent_d = ra.policy_gradient(full["pol"], full["traj"])
    g64, l64 = O.grad_f64_mt("policy", PS, pp, x, a.astype(np.uint8), adv)
    g32, _ = O.grad_f64_mt("policy", PS, pp, x, a.astype(np.uint8), adv, f32_samples=True)
    assert rel_err(g_d, g64) < max(4 * rel_err(g32, g64), 1e-6) and rel_err(g_d, g64) < 1e-5
    assert rel_err(g_d, g32) < 5e-6
    assert abs(loss_d - l64) <= 1e-6 * max(1.0, abs(l64))
    v = np.random.default_rng(7).standard_normal(len(pp)).astype(np.float32)
    h_d = ra.policy_fvp(full["pol"], full["traj"], v, 0.0)
    h64, _ = O.grad_f64_mt("fvp", PS, pp, x, v=v)
    assert rel_err(h_d, h64) < 1e-6
    gc_d, lc_d = ra.critic_gradient(full["cri"], full["traj"])
    gc64, lc64 = O.grad_f64_mt("critic", CS, cp, x, aux=rtg)
    assert rel_err(gc_d, gc64) < 1e-6
    assert abs(lc_d - lc64) <= 1e-6 * lc64


def test_critic_adam_steps_at_one_ranks_share_against_the_f32_oracle(engine):
    """five full-batch Adam steps of the critic on 8,192 x 128 samples (one rank's share of the headline at 8 GPUs)
    against oracle_critic_update_f32 — per-step losses and the parameters, tolerances of
    tests/test_gpu_parity.py::test_critic_gradient_and_update_match_oracle.  Reference: critics/opt.rs:100-126,
    torch/agents/mod.rs:35-72, optimizers/coptimizer.rs:158-167."""
    import ctypes as C
    n, steps = 8192, 5
    env = ra.CartPoleEnv(engine, n, max_steps=500, seed_env=0, seed_actor=1)
    pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
    pol.init(2)
    cri.init(3)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    x, _ = O.flat_samples(traj.read_all())
    rtg = np.ascontiguousarray(traj.read(ra.TRAJ_RETURNS).reshape(-1))
    cp = cri.get_params()
    st, losses_d = ra.critic_update(cri, ra.Adam(cri), traj, steps, want_losses=True)
    ad = O.lib().oracle_adam_new(len(cp))
    ac = O.AdamCfg()
    O.lib().oracle_adam_cfg_default(C.byref(ac))
    losses_o = np.zeros(steps, dtype=np.float32)
    c_o = cp.copy()
    O.lib().oracle_critic_update_f32(CS, O.f32p(c_o), ad, C.byref(ac), O.f32p(x), O.f32p(rtg), len(rtg), steps,
                                     O.f32p(losses_o))
    O.lib().oracle_adam_free(ad)
    assert np.allclose(losses_d, losses_o, rtol=1e-4)
    assert np.abs(cri.get_params() - c_o).max() < 2e-5 + 1e-3 * steps * 1e-3
    assert losses_d[-1] < losses_d[0] and st.steps == steps


def test_general_mlp_pair_kernel_at_size_against_f64(engine):
    """`[64, 64]` modules (k_gen_pair, kernels_gen_mfma.hip) at 16,384 x 128 = 2.1 M samples against an f64 NumPy
    restatement evaluated in chunks: surrogate gradient, Fisher-vector product, critic gradient; tolerances of
    tests/test_gpu_general_mlp.py::test_gradients_and_fisher_vector_products.  Reference: ff/mlp.rs:139-151."""
    from test_gpu_general_mlp import backward64, forward64, jvp64, unflatten
    n, hidden = 16384, [64, 64]
    pol, cri = ra.Mlp(engine, 5, hidden, 2), ra.Mlp(engine, 5, hidden, 1)
    pol.init(21)
    cri.init(22)
    env = ra.CartPoleEnv(engine, n, max_steps=500, seed_env=3, seed_actor=4)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    tr = traj.read_all()
    B = n * T
    x = np.ascontiguousarray(tr["obs"][:, :T, :].reshape(5, B).T)
    a = tr["action"].reshape(-1).astype(np.int64)
    adv = traj.read(ra.TRAJ_ADVANTAGES).reshape(-1).astype(np.float64)
    rtg = traj.read(ra.TRAJ_RETURNS).reshape(-1).astype(np.float64)
    pnet, cnet = unflatten(pol.get_params(), 5, hidden, 2), unflatten(cri.get_params(), 5, hidden, 1)
    vec = np.random.default_rng(5).normal(size=pol.P).astype(np.float32)
    tnet = unflatten(vec, 5, hidden, 2)
    want, fwant, cwant = np.zeros(pol.P), np.zeros(pol.P), np.zeros(cri.P)
    for lo in range(0, B, 1 << 17):
        sl = slice(lo, min(B, lo + (1 << 17)))
        z, acts = forward64(pnet, x[sl])
        zmax = z.max(axis=1, keepdims=True)
        lp = z - zmax - np.log(np.exp(z - zmax).sum(axis=1, keepdims=True))
        p = np.exp(lp)
        want += backward64(pnet, x[sl], acts, -(adv[sl, None]) * (np.eye(2)[a[sl]] - p) / B)
        _, tz = jvp64(pnet, tnet, x[sl])
        fwant += backward64(pnet, x[sl], acts, p * (tz - (p * tz).sum(axis=1, keepdims=True)) / B)
        v, cacts = forward64(cnet, x[sl])
        cwant += backward64(cnet, x[sl], cacts, 2.0 * (v - rtg[sl, None]) / B)
    got = ra.policy_gradient(pol, traj)[0]
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-9
    fgot = ra.policy_fvp(pol, traj, vec, 0.0)
    assert np.abs(fgot - fwant).max() <= 5e-5 * np.abs(fwant).max() + 1e-9
    cgot = ra.critic_gradient(cri, traj)[0]
    assert np.abs(cgot - cwant).max() <= 2e-5 * np.abs(cwant).max() + 1e-9


def test_full_size_return_scan_properties(engine, full):
    """gamma = 1 with unit rewards: the return is the number of steps to the end of the lane's episode segment;
    returns restart after every episode end; scaling gamma * lambda to 0 makes advantages one-step residuals"""
    traj = full["traj"]
    ra.reward_to_go(traj, 1.0)
    rtg = traj.read(ra.TRAJ_RETURNS)
    flag = full["data"]["flag"]
    ends = flag != 0
    ends[T - 1, :] = True
    # steps to the next end, computed with numpy on the full array
    idx = np.where(ends, np.arange(T)[:, None], T + 1)
    nxt = np.minimum.accumulate(idx[::-1], axis=0)[::-1]
    assert np.array_equal(rtg, (nxt - np.arange(T)[:, None] + 1).astype(np.float32))
    ra.gae(traj, full["cri"], 0.99, 0.0)
    adv0 = traj.read(ra.TRAJ_ADVANTAGES)
    v = full["values"]
    cont = ~ends
    delta = (1.0 + np.float32(0.99) * v[1:]) - v[:-1]
    assert np.array_equal(adv0[cont], delta.astype(np.float32)[cont])
    ra.gae(traj, full["cri"], 0.99, 0.95)  # restore for the tests below
    assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), full["adv"])


def test_full_size_update_obeys_the_trust_region(engine, full):
    pol, cri, traj = full["pol"], full["cri"], full["traj"]
    p0, c0 = pol.get_params().copy(), cri.get_params().copy()
    st = ra.trpo_update(pol, traj)
    assert st.status == ra.OPT_OK and st.cg_iterations == 10
    assert st.constraint_val_final <= 0.01 and st.loss_final < st.loss_initial
    loss_d, kl_d = ra.policy_loss_kl(pol, traj, p0)  # independent re-evaluation of what was accepted
    assert abs(loss_d - st.loss_final) <= 1e-5 * max(1.0, abs(st.loss_final))
    assert abs(kl_d - st.constraint_val_final) <= 1e-6 + 1e-3 * kl_d
    cs, losses = ra.critic_update(cri, ra.Adam(cri), traj, 80, want_losses=True)
    assert losses[-1] < losses[0] and np.all(np.isfinite(losses))
    pol.set_params(p0)
    cri.set_params(c0)


def test_config3_replay_invariants_and_samplers(engine):
    """configs[2]: 4,096 lanes, 50 M-step store (12,207 steps per lane), 100 k-step minibatches"""
    n, cap = 4096, 50_000_000 // 4096
    env = ra.CartPoleEnv(engine, n, max_steps=500, seed_env=0, seed_actor=1)
    q = ra.Mlp(engine, 5, H, 2)
    q.init(2)
    cfg = ra.dqn_config_default()
    cfg.buffer_capacity = cap
    cfg.exploration_kind, cfg.exploration_start = ra.SCHEDULE_CONSTANT, 0.5
    for i in range(8):
        cfg.agent_key[i] = 40 + i
    dqn = ra.Dqn(env, q, ra.Adam(q), cfg)
    total = 0
    for horizon in (5000, 5000, 4000):  # 14,000 > capacity: every lane evicts
        dqn.collect(horizon, want_stats=False)
        total += horizon
    head, count = dqn.replay_read(ra.REPLAY_HEAD).astype(np.int64), dqn.replay_read(ra.REPLAY_COUNT).astype(np.int64)
    eph, epc = dqn.replay_read(ra.REPLAY_EP_HEAD).astype(np.int64), dqn.replay_read(ra.REPLAY_EP_COUNT).astype(np.int64)
    assert np.all(dqn.replay_read(ra.REPLAY_TOTAL) == total) and np.all(count <= cap) and np.all(head > 0)
    assert np.all(head + count == total)  # nothing dangling: the horizon rule closes every episode
    ep_end, flag = dqn.replay_read(ra.REPLAY_EP_END), dqn.replay_read(ra.REPLAY_FLAG)
    for lane in (0, 1, 777, n - 1):
        ends = np.array([ep_end[(eph[lane] + k) % dqn.E, lane] for k in range(epc[lane])], dtype=np.int64)
        assert np.all(np.diff(ends) > 0) and ends[-1] == total and ends[0] > head[lane]
        # whole episodes only: the stored flags are non-zero exactly at the recorded episode ends
        slots = np.arange(head[lane], total) % cap
        nz = np.nonzero(flag[slots, lane])[0] + head[lane] + 1
        assert np.array_equal(nz, ends)
    pos0 = dqn.agent_rng_pos()
    ne, ns = dqn.minibatch_sample()
    lens = dqn.minibatch_read(ra.MB_EP_LEN).astype(np.int64)
    assert ns == lens.sum() >= 100_000 and ns - lens[-1] < 100_000
    assert dqn.agent_rng_pos() - pos0 == 2 * (ne + 1)  # one u64 per candidate, including the refused one
    lanes_par = dqn.minibatch_read(ra.MB_EP_LANE)
    assert np.array_equal(lanes_par, np.arange(ne) % n)  # the buffers are cycled in order
    tgt = dqn.minibatch_read(ra.MB_TARGET)
    off = dqn.minibatch_read(ra.MB_EP_OFFSET).astype(np.int64)
    # reward-to-go of unit rewards with gamma 0.99: the last step of every episode has target 1
    assert np.all(tgt[off + lens - 1] == 1.0) and np.all(np.diff(tgt[off[5]:off[5] + lens[5]]) < 0)
    st = dqn.update()
    assert np.isfinite(st.loss_last) and st.global_steps == total * n


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_config5_rollout_slices_and_update(engine, cell):
    """configs[4]: 16,384 Chain lanes under LatentStepLimit(100), GRU policy (and the same chain with the LSTM cell), T = 100"""
    n, Tc = 16384, 100
    gs = O.GruShape(5, 128, 128, 2) if cell == "gru" else O.LstmShape(5, 128, 128, 2)
    env = ra.ChainEnv(engine, n, max_steps=100, seed_env=3, seed_actor=4)
    Mod = ra.GruMlp if cell == "gru" else ra.LstmMlp
    pol, cri = Mod(engine, 5, 2), Mod(engine, 5, 1)
    pol.init(11)
    cri.init(12)
    traj = ra.Trajectory(engine, n, Tc, 5)
    ra.rollout(env, pol, traj)
    got = traj.read_all()
    for lo in (0, 9001, n - 32):
        sim = O.ChainLaneSim(32, max_steps=100, lane_offset=lo, seed_env=3, seed_actor=4)
        want = sim.rollout_gru(gs, pol.get_params(), Tc)
        for k in ("obs", "action", "reward", "flag"):
            assert np.array_equal(got[k][..., lo:lo + 32], want[k]), (lo, k)
    assert np.all(got["flag"][Tc - 1] == O.INTERRUPT) and not got["flag"][:Tc - 1].any()
    ra.gae(traj, cri, 0.95, 0.95)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 3
    st, losses = ra.ppo_update(pol, ra.Adam(pol), traj, cfg, want_losses=True)
    assert losses[-1] < losses[0] and 0.0 < st.entropy <= np.log(2.0) + 1e-6
    # the two builds of the training passes (recurrence on the bf16 matrix pipe with exact three-piece products — the
    # default — and the f32 kernels, engine kernel variant 1) agree at full size: 1.64 M sample-steps per gradient
    grads = {}
    for variant in (0, 1):
        engine.set_kernel_variant(variant)
        try:
            grads[variant] = (ra.policy_gradient(pol, traj)[0], ra.critic_gradient(cri, traj)[0])
        finally:
            engine.set_kernel_variant(0)
    for a, b in zip(grads[0], grads[1]):
        assert np.isfinite(a).all() and np.abs(a).max() > 0 and rel_err(a, b) < 1e-4, rel_err(a, b)


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_config5_gradients_against_the_f64_oracle_on_lane_slices(engine, cell):
    """The recurrent gradient passes AT config-5 size (16,384 lanes x 100 steps: 512 tiles, 51,200 (step, tile) blocks,
    every chunk and flush path of the five kernels) against the f64 BPTT oracle.  A gradient is a sum over lanes, and the
    f64 oracle over 1.64 M sample-steps would take minutes, so the check uses linearity: the advantages (policy) /
    regression residuals (critic) are made zero on every lane outside three 32-lane slices — first tile, a slice that
    straddles two tiles in the middle, last tile — and the device's full-size gradient must then equal the oracle's
    gradient of those 96 lanes, weighted by their share of the batch.  (The bf16-pipe build against the f32 kernels at
    this size is the test above; this one pins both to the oracle.)  Reference: seq/rnn/gru.rs:20-98, lstm.rs,
    modules/chain.rs:163-186, policies/trpo.rs:124-131, critics/opt.rs:109-115."""
    n, Tc = 16384, 100
    gs_p = O.GruShape(5, 128, 128, 2) if cell == "gru" else O.LstmShape(5, 128, 128, 2)
    gs_c = O.GruShape(5, 128, 128, 1) if cell == "gru" else O.LstmShape(5, 128, 128, 1)
    env = ra.ChainEnv(engine, n, max_steps=100, seed_env=3, seed_actor=4)
    Mod = ra.GruMlp if cell == "gru" else ra.LstmMlp
    pol, cri = Mod(engine, 5, 2), Mod(engine, 5, 1)
    pol.init(11)
    cri.init(12)
    traj = ra.Trajectory(engine, n, Tc, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.95, 0.95)
    got = traj.read_all()
    adv, rtg = traj.read(ra.TRAJ_ADVANTAGES), traj.read(ra.TRAJ_RETURNS)
    slices = (0, 9001, n - 32)
    keep = np.zeros(n, dtype=bool)
    for lo in slices:
        keep[lo:lo + 32] = True
    B = n * Tc

    def lanes(lo):
        return dict(obs=np.ascontiguousarray(got["obs"][..., lo:lo + 32]), action=np.ascontiguousarray(got["action"][:, lo:lo + 32]),
                    flag=np.ascontiguousarray(got["flag"][:, lo:lo + 32]),
                    term_obs=np.ascontiguousarray(got["term_obs"][..., lo:lo + 32]))

    # ---- policy: zero advantages outside the slices (their d loss / d logits is then exactly 0)
    traj.write(ra.TRAJ_ADVANTAGES, np.where(keep[None, :], adv, 0.0).astype(np.float32))
    g_d, _, _ = ra.policy_gradient(pol, traj)
    p = pol.get_params()
    g64 = np.zeros(len(p))
    for lo in slices:
        want = lanes(lo)
        logits, _ = O.gru_seq_forward(gs_p, p, want, f64=True, want_succ=False)
        z = logits - logits.max(0)
        pr = np.exp(z - np.log(np.exp(z).sum(0)))
        a = want["action"].astype(np.int64)
        ind = np.stack([a == 0, a == 1]).astype(np.float64)
        dl = -(adv[:, lo:lo + 32].astype(np.float64) / B) * (ind - pr)
        g64 += O.gru_seq_backward(gs_p, p, want, dl, f64=True)
    assert np.abs(g64).max() > 0 and rel_err(g_d, g64) < 5e-6, rel_err(g_d, g64)
    # ---- critic: outside the slices the targets are the critic's own values (residual 0 up to the last bit of two f32
    # forwards), inside them the returns
    values, _ = cri.seq_forward(traj, want_succ=False)
    traj.write(ra.TRAJ_RETURNS, np.where(keep[None, :], rtg, values[0]).astype(np.float32))
    gc_d, _ = ra.critic_gradient(cri, traj)
    pc = cri.get_params()
    gc64 = np.zeros(len(pc))
    for lo in slices:
        want = lanes(lo)
        v, _ = O.gru_seq_forward(gs_c, pc, want, f64=True, want_succ=False)
        d = v - rtg[:, lo:lo + 32].astype(np.float64)[None]
        gc64 += O.gru_seq_backward(gs_c, pc, want, 2.0 * d / B, f64=True)
    assert np.abs(gc64).max() > 0 and rel_err(gc_d, gc64) < 5e-6, rel_err(gc_d, gc64)
