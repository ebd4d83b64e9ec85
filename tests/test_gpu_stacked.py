"""Recurrent chains with RnnBaseConfig::num_layers > 1 (src/torch/modules/seq/rnn/mod.rs:20-45,223-257) on the
lane-per-thread kernels (relearn_amd/csrc/kernels_seq_stack.hip): initialisation and forward bit for bit against the C
restatement (oracle/stack_impl.inc), gradients / Fisher-vector products / updates against the NumPy f64 restatement
(oracle/stacked.py) — both pinned by the PyTorch vectors of tests/golden/torch_golden_stacked.json — rollouts on either
lane family, the actor documents, and the PyTorch vectors themselves through the device."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O
import relearn_amd as ra
from oracle import stacked as S

pytestmark = pytest.mark.gpu
L = O.lib()

# (cell, in_dim, hidden, num_layers, mlp_hidden, out_dim): widths that are / are not multiples of the unit quads
SHAPES = [("gru", 5, 32, 2, 16, 2), ("lstm", 3, 18, 3, 10, 1), ("gru", 2, 7, 4, 5, 1), ("lstm", 5, 64, 2, 33, 2),
          ("gru", 5, 128, 2, 128, 2),
          # more input features than the fused tile kernels' five: the same lane-per-thread kernels, one layer or several
          ("gru", 7, 12, 1, 8, 2), ("lstm", 8, 16, 2, 8, 1),
          # wider than the fused kernels' 128
          ("gru", 5, 160, 1, 130, 2), ("lstm", 4, 132, 2, 40, 1)]
GRAD_RTOL = 5e-6


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def modules(engine, cell, D, H, NL, H2, A, seed):
    cls = ra.GruMlp if cell == "gru" else ra.LstmMlp
    m = cls(engine, D, A, H, H2, num_layers=NL)
    m.init(seed)
    shape = O.GruShape(D, H, H2, A, O.CELL_GRU if cell == "gru" else O.CELL_LSTM)
    spec = S.Spec(S.GRU if cell == "gru" else S.LSTM, D, H, NL, H2, A)
    assert m.P == L.oracle_stack_num_params(shape, NL) == spec.num_params()
    assert np.array_equal(m.get_params(), O.stack_init(shape, NL, seed))  # RnnWeights::new's layer loop, bit for bit
    return m, shape, spec


def synthetic_history(engine, n, T, D, seed):
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


@pytest.mark.parametrize("cell,D,H,NL,H2,A", SHAPES)
@pytest.mark.parametrize("n", [70, 128])
def test_forward_is_bit_exact(engine, cell, D, H, NL, H2, A, n):
    """module outputs and successor outputs (Interrupt and horizon cuts) against the C restatement: identical bits; any
    lane count (no tile condition on this path)"""
    m, shape, _ = modules(engine, cell, D, H, NL, H2, A, 31)
    traj, want = synthetic_history(engine, n, 11, D, 5)
    out_d, succ_d = m.seq_forward(traj)
    out_o, succ_o = O.stack_seq_forward(shape, NL, m.get_params(), want)
    assert np.array_equal(out_d, out_o) and np.array_equal(succ_d, succ_o)
    assert np.abs(out_o).max() > 0 and np.count_nonzero(succ_o) > 0
    if A == 1:  # values, advantages and returns through the recurrent critic path
        ra.gae(traj, m, 0.95, 0.9)
        adv_o, rtg_o = O.seq_gae(out_o[0], succ_o[0], want, np.float32(0.95), np.float32(0.9))
        assert np.array_equal(traj.read(ra.TRAJ_ADVANTAGES), adv_o) and np.array_equal(traj.read(ra.TRAJ_RETURNS), rtg_o)


@pytest.mark.parametrize("cell,D,H,NL,H2,A", SHAPES)
def test_gradients_against_f64(engine, cell, D, H, NL, H2, A):
    """surrogate / critic gradient through time and (policies) a Fisher-vector product against the f64 restatement; every
    parameter block of every layer receives gradient"""
    m, shape, spec = modules(engine, cell, D, H, NL, H2, A, 41)
    traj, want = synthetic_history(engine, 96, 9, D, 6)
    p = m.get_params()
    B = want["action"].size
    if A == 2:
        g_d, loss_d, ent_d = ra.policy_gradient(m, traj)
        logits, _, _ = S.forward(spec, p, want, want_succ=False)
        z = logits - logits.max(0)
        lp = z - np.log(np.exp(z).sum(0))
        pr = np.exp(lp)
        a = want["action"].astype(np.int64)
        ind = np.stack([a == 0, a == 1]).astype(np.float64)
        g64 = S.backward(spec, p, want, -(want["adv"].astype(np.float64) / B) * (ind - pr))
        assert g_d.shape == g64.shape and rel_err(g_d, g64) < GRAD_RTOL, rel_err(g_d, g64)
        assert abs(loss_d + want["adv"].astype(np.float64).mean()) < 1e-6  # -mean(ratio * A) at ratio 1
        v = np.random.default_rng(3).normal(size=m.P).astype(np.float32)
        h_d = ra.policy_fvp(m, traj, v, 1e-5)
        h64 = S.policy_fvp(spec, p, v, want, 1e-5)
        assert rel_err(h_d, h64) < 2e-5, rel_err(h_d, h64)
    else:
        g_d, loss_d = ra.critic_gradient(m, traj)
        v, _, _ = S.forward(spec, p, want, want_succ=False)
        d = v - want["rtg"].astype(np.float64)[None]
        g64 = S.backward(spec, p, want, 2.0 * d / B)
        assert rel_err(g_d, g64) < GRAD_RTOL, rel_err(g_d, g64)
        assert abs(loss_d - (d * d).mean()) <= 1e-5 * (d * d).mean()
        st, losses = ra.critic_update(m, ra.Adam(m), traj, 3, want_losses=True)
        assert losses[-1] < losses[0] and not np.array_equal(m.get_params(), p)
    for name, l, shp, o in spec.slices()[0]:
        if name in ("bih", "bhh", "Wih", "Whh", "W1", "W2"):
            blk = slice(o, o + int(np.prod(shp)))
            assert np.abs(g_d[blk]).max() > 0, (name, l)
            assert rel_err(g_d[blk], g64[blk]) < 100 * GRAD_RTOL, (name, l, rel_err(g_d[blk], g64[blk]))


def test_torch_vectors_through_the_device(engine):
    """the committed PyTorch vectors (f32 cases) fed through the device: outputs, successor outputs, gradient of
    sum(dout * out) via the critic-gradient seam is not available for arbitrary dout, so outputs and the Fisher / gradient
    pieces are covered above — here the forward at torch's own parameters and histories"""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "torch_golden_stacked.json")) as f:
        gold = json.load(f)
    for name in ("gru_l2_f32", "lstm_l2_f32"):
        c = gold[name]
        D, H, NL, H2, A = c["dims"]
        n, T = c["n"], c["T"]
        cls = ra.GruMlp if c["cell"] == "gru" else ra.LstmMlp
        m = cls(engine, D, A, H, H2, num_layers=NL)
        m.set_params(np.array(c["params"], dtype=np.float32))
        rng = np.random.default_rng(0)
        want = {"obs": np.array(c["obs"], dtype=np.float32).reshape(D, T + 1, n),
                "term_obs": np.array(c["term_obs"], dtype=np.float32).reshape(D, T, n),
                "flag": np.array(c["flag"], dtype=np.uint8).reshape(T, n),
                "action": rng.integers(0, 2, size=(T, n)).astype(np.uint8),
                "reward": np.zeros((T, n), dtype=np.float32)}
        traj = ra.Trajectory(engine, n, T, D)
        traj.write_all(want)
        out_d, succ_d = m.seq_forward(traj)
        assert np.allclose(out_d, np.array(c["out"]).reshape(A, T, n), rtol=2e-5, atol=2e-6)
        assert np.allclose(succ_d, np.array(c["succ_out"]).reshape(A, T, n), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_rollout_on_cartpole_lanes(engine, cell):
    """the env side replays bit for bit through the oracle's lanes; the recorded actions are the inverse-CDF draws of the
    policy's distribution at the logits of the teacher-forced forward (states restart at episode ends and at the start of
    a collection), which the rollout's step kernel reproduces bit for bit; a second collection continues the lanes"""
    n, T = 96, 14
    pol, shape, _ = modules(engine, cell, 5, 24, 2, 12, 2, 11)
    env = ra.CartPoleEnv(engine, n, max_steps=9, seed_env=3, seed_actor=4)
    traj = ra.Trajectory(engine, n, T, 5)
    sim = O.LaneSim(n, max_steps=9, seed_env=3, seed_actor=4)
    r = O.Prng()
    for period in range(2):
        ra.rollout(env, pol, traj)
        got = traj.read_all()
        assert np.array_equal(got["obs"][:, 0, :], sim.observe())
        for t in range(T):
            reward, flag, obs, term = sim.step(got["action"][t])
            assert np.array_equal(got["reward"][t], reward) and np.array_equal(got["flag"][t], flag), t
            assert np.array_equal(got["obs"][:, t + 1, :], obs), t
            msk = flag == O.INTERRUPT
            assert np.array_equal(got["term_obs"][:, t, msk], term[:, msk])
        assert (got["flag"] != O.CONTINUE).any()
        z, _ = O.stack_seq_forward(shape, 2, pol.get_params(), got, want_succ=False)
        zd, _ = pol.seq_forward(traj, want_succ=False)
        assert np.array_equal(z, zd)
        checked = 0
        for t in range(T):
            p0 = 1.0 / (1.0 + np.exp(z[1, t].astype(np.float64) - z[0, t]))
            for i in range(n):
                L.oracle_prng_seed_from_u64(C.byref(r), 4)
                L.oracle_prng_set_stream(C.byref(r), i)
                L.oracle_prng_set_word_pos(C.byref(r), period * T + t)
                u = L.oracle_prng_gen_f32(C.byref(r))
                if abs(u - p0[i]) > 1e-6:
                    assert got["action"][t, i] == (0 if u < p0[i] else 1), (period, t, i)
                    checked += 1
        assert checked > 0.99 * n * T


def test_rollout_and_updates_on_chain_lanes(engine):
    """the index-env lanes (Chain): env side against the oracle's lanes, then one update of each kind lowers its loss —
    PPO, REINFORCE-style policy gradient steps, critic fitting, TRPO"""
    n, T = 64, 24
    env = ra.ChainEnv(engine, n, max_steps=9, seed_env=3, seed_actor=4)
    sim = O.ChainLaneSim(n, max_steps=9, seed_env=3, seed_actor=4)
    pol, _, _ = modules(engine, "gru", 5, 20, 2, 16, 2, 13)
    cri, _, _ = modules(engine, "lstm", 5, 16, 2, 8, 1, 14)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    got = traj.read_all()
    assert np.array_equal(got["obs"][:, 0, :], sim.observe())
    for t in range(T):
        reward, flag, obs, term = sim.step(got["action"][t])
        assert np.array_equal(got["reward"][t], reward) and np.array_equal(got["flag"][t], flag), t
        assert np.array_equal(got["obs"][:, t + 1, :], obs), t
    ra.gae(traj, cri, 0.95, 0.9)
    st, losses = ra.critic_update(cri, ra.Adam(cri), traj, 4, want_losses=True)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    p0 = pol.get_params()
    st = ra.trpo_update(pol, traj)
    assert st.status == ra.OPT_OK and st.constraint_val_final <= 0.01 and st.loss_final < st.loss_initial
    assert not np.array_equal(pol.get_params(), p0)
    cfg = ra.ppo_config_default()
    cfg.opt_steps_per_update = 4
    st, losses = ra.ppo_update(pol, ra.Adam(pol), traj, cfg, want_losses=True)
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_trpo_step_against_the_f64_pieces(engine):
    """one TRPO update of a stacked policy: the step the device takes satisfies the f64 restatement's view of it — the
    surrogate loss falls and the mean KL (recomputed in f64 from the two parameter vectors) matches the reported one"""
    m, shape, spec = modules(engine, "gru", 5, 16, 2, 12, 2, 17)
    traj, want = synthetic_history(engine, 64, 10, 5, 8)
    p0 = m.get_params()
    st = ra.trpo_update(m, traj)
    p1 = m.get_params()
    assert st.status == ra.OPT_OK and not np.array_equal(p0, p1)
    l0, _, _ = S.forward(spec, p0, want, want_succ=False)
    l1, _, _ = S.forward(spec, p1, want, want_succ=False)
    lp0, lp1 = l0 - np.log(np.exp(l0).sum(0)), l1 - np.log(np.exp(l1).sum(0))
    a = want["action"].astype(np.int64)
    sel = lambda lp: np.where(a == 0, lp[0], lp[1])
    loss_o = -(np.exp(sel(lp1) - sel(lp0)) * want["adv"]).mean()
    kl_o = (np.exp(lp0) * (lp0 - lp1)).sum(0).mean()
    assert abs(st.loss_final - loss_o) <= 2e-5 * max(1.0, abs(loss_o))
    assert abs(st.constraint_val_final - kl_o) <= 1e-3 * kl_o + 3e-8 and 1e-5 < kl_o <= 0.0101


@pytest.mark.parametrize("kind", ["stacked", "single-layer", "general"])
def test_actor_critic_update_with_modules_off_the_fused_path(engine, kind):
    """rl_actor_critic_update (the batch_update verb) with modules the two-stream pairing is not built for — stacked and
    single-layer recurrent chains, an MLP on the per-layer kernels — runs the two updates in turn on one stream:
    bit-identical to rl_trpo_update followed by rl_values_opt_update"""
    n, T = 64, 20
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update = 5

    def run(joint):
        env = ra.CartPoleEnv(engine, n, max_steps=9, seed_env=5, seed_actor=6)
        if kind == "stacked":
            pol, cri = ra.GruMlp(engine, 5, 2, 16, 12, num_layers=2), ra.LstmMlp(engine, 5, 1, 12, 8, num_layers=2)
        elif kind == "single-layer":
            pol, cri = ra.GruMlp(engine, 5, 2, 32, 16), ra.GruMlp(engine, 5, 1, 32, 16)
        else:
            pol, cri = ra.Mlp(engine, 5, [24, 24], 2), ra.Mlp(engine, 5, [130], 1)
        pol.init(2)
        cri.init(3)
        opt = ra.Adam(cri)
        traj = ra.Trajectory(engine, n, T, 5)
        ra.rollout(env, pol, traj)
        ra.gae(traj, cri, 0.99, 0.95)
        if joint:
            pst, cst, losses = ra.actor_critic_update(pol, cri, opt, traj, None, ccfg, want_losses=True)
        else:
            pst = ra.trpo_update(pol, traj)
            cst, losses = ra.values_opt_update(cri, opt, traj, ccfg, want_losses=True)
        return pol.get_params(), cri.get_params(), pst.as_dict(), losses.copy()

    a, b = run(False), run(True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])
    assert a[2]["status"] == ra.OPT_OK and a[3][-1] < a[3][0]


def test_actor_document_round_trip(engine):
    """RnnWeights { flat_weights } holds 4 tensors per layer (seq/rnn/mod.rs:186-191,223-257): written and read back"""
    pol, _, _ = modules(engine, "lstm", 5, 12, 3, 6, 2, 19)
    env = ra.CartPoleEnv(engine, 64, max_steps=9, seed_env=3, seed_actor=4)
    doc = ra.actor_to_cbor(env, pol)
    p = pol.get_params()
    other = ra.LstmMlp(engine, 5, 2, 12, 6, num_layers=3)
    ra.module_from_cbor(other, doc)
    assert np.array_equal(other.get_params(), p)
    two = ra.LstmMlp(engine, 5, 2, 12, 6, num_layers=2)
    with pytest.raises(ra.RelearnError):
        ra.module_from_cbor(two, doc)


def test_layer_counts_that_are_not_built_are_refused(engine):
    h = C.c_void_p()
    assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(0), C.c_uint32(5), C.c_uint32(32), C.c_uint32(5),
                                      C.c_uint32(32), C.c_uint32(2), C.byref(h)) == ra.ERR_UNSUPPORTED
    assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(0), C.c_uint32(5), C.c_uint32(32), C.c_uint32(0),
                                      C.c_uint32(32), C.c_uint32(2), C.byref(h)) == ra.ERR_BUILD_AGENT
    assert ra.lib().rl_rnn_mlp_create(engine.h, C.c_int32(1), C.c_uint32(9), C.c_uint32(32), C.c_uint32(2),
                                      C.c_uint32(32), C.c_uint32(2), C.byref(h)) == ra.ERR_BUILD_AGENT


def test_init_with_initializers(engine):
    """rl_rnn_mlp_init_with: RnnBaseConfig { input_weights_init, hidden_weights_init, bias_init } and the chain MLP's
    LinearConfig, against the oracle's stream bit for bit; bias_init = None and Orthogonal biases are refused"""
    inits = (("Normal", "FanIn", 0.0), ("Uniform", "Constant", 0.25), ("Normal", "FanOut", 0.0), ("Orthogonal", "FanAvg", 0.0),
             ("Constant", "FanAvg", 0.5))
    for cell, NL in (("lstm", 2), ("gru", 1), ("gru", 3)):
        cls = ra.GruMlp if cell == "gru" else ra.LstmMlp
        m = cls(engine, 5, 2, 16, 8, num_layers=NL)
        shape = O.GruShape(5, 16, 8, 2, O.CELL_GRU if cell == "gru" else O.CELL_LSTM)
        m.init_with(7, *inits)
        assert np.array_equal(m.get_params(), O.stack_init_with(shape, NL, 7, inits))
        m.init_with(9)  # the defaults
        assert np.array_equal(m.get_params(), O.stack_init(shape, NL, 9))
        with pytest.raises(ra.RelearnError) as e:  # (bias_init = None belongs to modules built without bias vectors)
            m.init_with(7, bias=None)
        assert e.value.code == ra.ERR_INVALID_ARGUMENT
        with pytest.raises(ra.RelearnError) as e:
            m.init_with(7, mlp_bias=None)
        assert e.value.code == ra.ERR_UNSUPPORTED
        with pytest.raises(ra.RelearnError) as e:
            m.init_with(7, bias=("Orthogonal", "FanAvg", 0.0))
        assert e.value.code == ra.ERR_INVALID_ARGUMENT
    with pytest.raises(ra.RelearnError):
        ra.lib().rl_rnn_mlp_init_with  # (exported)
        mlp = ra.Mlp(engine, 5, 32, 2)
        ra._check(ra.lib().rl_rnn_mlp_init_with(mlp.h, C.c_uint64(1), None, None, None, None, None), engine.h)


def _bias_index(spec):
    """indices of the recurrent bias entries in the flat vector of the module WITH bias vectors"""
    idx = []
    for name, l, shp, o in spec.slices()[0]:
        if name in ("bih", "bhh"):
            idx += list(range(o, o + int(np.prod(shp))))
    return np.array(idx)


@pytest.mark.parametrize("cell,D,H,NL,H2,A", [("gru", 5, 128, 1, 128, 2), ("lstm", 5, 24, 2, 12, 2), ("gru", 3, 10, 3, 6, 1)])
def test_recurrent_layers_without_bias_vectors(engine, cell, D, H, NL, H2, A):
    """RnnBaseConfig::bias_init = None (seq/rnn/mod.rs:20-45,246-251): the flat vector holds [W_ih, W_hh] per layer; the
    default initialisation draws the same stream (Zeros draw nothing); outputs are bit-identical to the module with zero
    bias vectors, gradients and Fisher-vector products are that module's without the bias entries (f64 restatement);
    actor documents carry has_biases: false"""
    cls = ra.GruMlp if cell == "gru" else ra.LstmMlp
    m = cls(engine, D, A, H, H2, num_layers=NL, rnn_bias=False)
    full, shape, spec = modules(engine, cell, D, H, NL, H2, A, 51)
    keep = np.setdiff1d(np.arange(full.P), _bias_index(spec))
    assert m.P == len(keep)
    m.init(51)
    assert np.array_equal(m.get_params(), full.get_params()[keep])
    traj, want = synthetic_history(engine, 70, 8, D, 9)
    out_b, succ_b = m.seq_forward(traj)
    out_f, succ_f = O.stack_seq_forward(shape, NL, full.get_params(), want)
    assert np.array_equal(out_b, out_f) and np.array_equal(succ_b, succ_f)
    p = full.get_params()
    B = want["action"].size
    if A == 2:
        g_d = ra.policy_gradient(m, traj)[0]
        logits, _, _ = S.forward(spec, p, want, want_succ=False)
        z = logits - logits.max(0)
        pr = np.exp(z - np.log(np.exp(z).sum(0)))
        a = want["action"].astype(np.int64)
        ind = np.stack([a == 0, a == 1]).astype(np.float64)
        g64 = S.backward(spec, p, want, -(want["adv"].astype(np.float64) / B) * (ind - pr))[keep]
        assert rel_err(g_d, g64) < GRAD_RTOL
        v = np.random.default_rng(3).normal(size=m.P).astype(np.float32)
        vfull = np.zeros(full.P)
        vfull[keep] = v
        h64 = S.policy_fvp(spec, p, vfull, want, 1e-5)[keep]
        assert rel_err(ra.policy_fvp(m, traj, v, 1e-5), h64) < 2e-5
        if D == 5:  # rolls out on an env and learns
            env = ra.ChainEnv(engine, 64, max_steps=9, seed_env=3, seed_actor=4)
            t2 = ra.Trajectory(engine, 64, 16, 5)
            ra.rollout(env, m, t2)
            ra.reward_to_go(t2, 0.95)
            cfg = ra.ppo_config_default()
            cfg.opt_steps_per_update = 3
            st, losses = ra.ppo_update(m, ra.Adam(m), t2, cfg, want_losses=True)
            assert np.all(np.isfinite(losses)) and losses[-1] < losses[0]
    else:
        g_d, loss_d = ra.critic_gradient(m, traj)
        vv, _, _ = S.forward(spec, p, want, want_succ=False)
        d = vv - want["rtg"].astype(np.float64)[None]
        assert rel_err(g_d, S.backward(spec, p, want, 2.0 * d / B)[keep]) < GRAD_RTOL
    env = ra.CartPoleEnv(engine, 64, max_steps=9, seed_env=3, seed_actor=4)
    if D == 5 and A == 2:
        doc = ra.actor_to_cbor(env, m)
        other = cls(engine, D, A, H, H2, num_layers=NL, rnn_bias=False)
        ra.module_from_cbor(other, doc)
        assert np.array_equal(other.get_params(), m.get_params())
        with pytest.raises(ra.RelearnError):
            ra.module_from_cbor(full, doc)
    m.init_with(5, bias=None)
    assert m.get_params().shape == (m.P,)
