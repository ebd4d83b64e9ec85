"""MLPs of any `MlpConfig.hidden_sizes` (src/torch/modules/ff/mlp.rs:13-34) through the C ABI: shapes the fused
single-hidden-layer kernels do not cover run the per-layer kernels of relearn_amd/csrc/kernels_general.hip.  The checker
here is an f64 NumPy restatement of the same network (forward, backward, forward-mode tangents) around the oracle's
array-fed pieces (loss terms from logits, GAE / TD targets from value arrays); the env side of the step-by-step rollout
is replayed through the oracle's lanes."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
import relearn_amd as ra

pytestmark = pytest.mark.gpu

SHAPES = [[], [64, 64], [256], [32, 16, 8], [130], [200, 3, 256, 17]]


def layers(in_dim, hidden, out_dim):
    w = [in_dim] + list(hidden) + [out_dim]
    return list(zip(w[:-1], w[1:]))


def unflatten(p, in_dim, hidden, out_dim):
    out, k = [], 0
    for fi, fo in layers(in_dim, hidden, out_dim):
        W = p[k:k + fi * fo].reshape(fo, fi)
        k += fi * fo
        b = p[k:k + fo]
        k += fo
        out.append((W.astype(np.float64), b.astype(np.float64)))
    assert k == p.size
    return out


def act64(name, pre):
    """Activation::forward (ff/activation.rs:85-92) -> (output, derivative)"""
    if name == "Relu":
        return np.maximum(pre, 0.0), (pre > 0).astype(np.float64)
    if name == "Sigmoid":
        y = 1.0 / (1.0 + np.exp(-pre))
        return y, y * (1.0 - y)
    if name == "Tanh":
        y = np.tanh(pre)
        return y, 1.0 - y * y
    return pre, np.ones_like(pre)


def forward64(net, x, act="Relu", out_act="Identity"):
    """x [rows][in] -> (outputs [rows][out], per layer (layer outputs, slopes))"""
    acts, h = [], x.astype(np.float64)
    for i, (W, b) in enumerate(net):
        h, slope = act64(act if i + 1 < len(net) else out_act, h @ W.T + b)
        acts.append((h, slope))
    return h, acts


def backward64(net, x, acts, dz):
    """gradient of sum(dz * outputs) w.r.t. the flat parameters"""
    grads, d = [], dz * acts[-1][1]
    inputs = [x.astype(np.float64)] + [h for h, _ in acts[:-1]]
    for i in reversed(range(len(net))):
        W, _ = net[i]
        grads.append((d.T @ inputs[i], d.sum(axis=0)))
        if i > 0:
            d = (d @ W) * acts[i - 1][1]
    return np.concatenate([np.concatenate([gw.ravel(), gb]) for gw, gb in reversed(grads)])


def jvp64(net, tnet, x, act="Relu", out_act="Identity"):
    h, th = x.astype(np.float64), np.zeros_like(x, dtype=np.float64)
    for i, ((W, b), (V, vb)) in enumerate(zip(net, tnet)):
        th = th @ W.T + h @ V.T + vb
        h, slope = act64(act if i + 1 < len(net) else out_act, h @ W.T + b)
        th = th * slope
    return h, th


def make(engine, in_dim, hidden, out_dim, seed, act="Relu", out_act="Identity"):
    m = ra.Mlp(engine, in_dim, hidden, out_dim, act, out_act)
    m.init(seed)
    return m


@pytest.mark.parametrize("hidden", SHAPES)
def test_shapes_init_and_forward(engine, hidden):
    rng = np.random.default_rng(1)
    for in_dim, out_dim in ((5, 2), (4, 1)):
        m = make(engine, in_dim, hidden, out_dim, 7)
        assert m.P == sum(fi * fo + fo for fi, fo in layers(in_dim, hidden, out_dim))
        p = m.get_params()
        k = 0
        for fi, fo in layers(in_dim, hidden, out_dim):  # Linear::new: Uniform(+-sqrt(6 / (fan_in + 1 + fan_out)))
            lim = np.float32(np.sqrt(3.0 * (2.0 / ((fi + 1) + fo))))
            blk = p[k:k + fi * fo + fo]
            k += fi * fo + fo
            assert np.abs(blk).max() <= lim and (blk.size < 50 or np.abs(blk).max() > 0.8 * lim)
        assert np.array_equal(p, make(engine, in_dim, hidden, out_dim, 7).get_params())
        x = rng.normal(size=(300, in_dim)).astype(np.float32)
        want, _ = forward64(unflatten(p, in_dim, hidden, out_dim), x)
        got = m.forward(x)
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("hidden,act,out_act", [([64, 64], "Tanh", "Identity"), ([32, 16, 8], "Sigmoid", "Tanh"),
                                                ([128], "Sigmoid", "Identity"), ([100], "Identity", "Sigmoid"),
                                                ([], "Relu", "Tanh"), ([256, 3], "Relu", "Identity")])
def test_forward_with_activations_is_bit_exact_against_the_oracle(engine, hidden, act, out_act):
    """the per-layer kernels' forward — fma order, the shared deterministic sigmoid / tanh — reproduces
    oracle_mlp_layers_forward_f32 (Mlp::forward, ff/mlp.rs:139-151) bit for bit"""
    rng = np.random.default_rng(4)
    for in_dim, out_dim in ((5, 2), (4, 1)):
        m = make(engine, in_dim, hidden, out_dim, 9, act, out_act)
        x = (3.0 * rng.normal(size=(257, in_dim))).astype(np.float32)
        want = O.mlp_layers_forward(in_dim, hidden, out_dim, m.get_params(), x, act, out_act)
        assert np.array_equal(m.forward(x), want)
        ref, _ = forward64(unflatten(m.get_params(), in_dim, hidden, out_dim), x, act, out_act)
        assert np.allclose(want, ref, rtol=2e-5, atol=2e-6)  # and the oracle agrees with the f64 restatement


def test_actor_documents_carry_the_activation_variants(engine):
    """Mlp { layers, activation, output_activation } (ff/mlp.rs:45-50): unit variants as serde writes them — their names"""
    from cbor_ref import decode
    env = ra.CartPoleEnv(engine, 64)
    for act, out_act in (("Tanh", "Identity"), ("Sigmoid", "Tanh"), ("Relu", "Identity")):
        pol = make(engine, 5, [24, 12], 2, 5, act, out_act)
        doc = ra.actor_to_cbor(env, pol)
        def find(v):  # the Mlp map inside the actor document
            if isinstance(v, dict):
                if "activation" in v and "layers" in v:
                    return v
                v = list(v.values())
            if isinstance(v, (list, tuple)):
                for item in v:
                    hit = find(item)
                    if hit is not None:
                        return hit
            return None
        mod = find(decode(bytes(doc)))
        assert mod is not None and mod["activation"] == act and mod["output_activation"] == out_act
        assert len(mod["layers"]) == 3
        twin = ra.Mlp(engine, 5, [24, 12], 2, act, out_act)
        ra.module_from_cbor(twin, doc)
        assert np.array_equal(twin.get_params(), pol.get_params())
        other = ra.Mlp(engine, 5, [24, 12], 2, "Relu" if act != "Relu" else "Tanh", out_act)
        with pytest.raises(ra.RelearnError):  # a document of other activations is not this module's
            ra.module_from_cbor(other, doc)


def test_invalid_activations_are_refused(engine):
    sizes = (C.c_uint32 * 1)(64)
    h = C.c_void_p()
    for a, o in ((4, 0), (-1, 0), (1, 7)):
        code = ra.lib().rl_mlp_create_layers(engine.h, C.c_uint32(5), sizes, C.c_uint32(1), C.c_uint32(2), C.c_int32(a),
                                             C.c_int32(o), C.byref(h))
        assert code == ra.ERR_BUILD_AGENT and not h
    # the reference's defaults on one hidden layer of <= 128 units are the fused module, anything else the general one
    q = ra.Mlp(engine, 5, [64], 2, "Tanh", "Identity")
    env = ra.CartPoleEnv(engine, 64)
    cfg = ra.dqn_config_default()
    cfg.buffer_capacity, cfg.minibatch_steps = 64, 100
    ra.Dqn(env, q, ra.Adam(q), cfg).close()  # (DQN takes any feed-forward module: tests/test_gpu_dqn.py)


def test_unsupported_shapes_are_refused(engine):
    for hidden in ([300], [8] * 5, [0]):
        with pytest.raises(ra.RelearnError):
            ra.Mlp(engine, 5, hidden, 2)
    for in_dim in (0, 9):
        with pytest.raises(ra.RelearnError):
            ra.Mlp(engine, in_dim, [32, 32], 2)
    q = ra.GruMlp(engine, 5, 2)
    env = ra.CartPoleEnv(engine, 64)
    with pytest.raises(ra.RelearnError):  # DQN: feed-forward action-value modules only
        ra.Dqn(env, q, ra.Adam(q), ra.dqn_config_default())


@pytest.mark.parametrize("in_dim,hidden,n,T", [(3, [16, 16], 70, 9), (7, [32], 64, 6), (8, [130], 33, 5), (1, [], 96, 4),
                                               (6, [64, 64], 128, 8)])
def test_other_input_widths_on_host_made_histories(engine, in_dim, hidden, n, T):
    """MlpConfig modules of in_dim 1..8 (the envs here have 4 or 5 features: the observations come from rl_traj_write):
    values / GAE, the surrogate and critic gradients and a Fisher-vector product against the f64 NumPy network — the
    matrix-pipe launch takes up to seven inputs, the per-layer kernels any"""
    rng = np.random.default_rng(in_dim * 31 + len(hidden))
    pol, cri = make(engine, in_dim, hidden, 2, 21), make(engine, in_dim, hidden, 1, 22)
    want = {"obs": rng.normal(size=(in_dim, T + 1, n)).astype(np.float32),
            "flag": rng.choice(np.array([0, 0, 0, 1, 2], dtype=np.uint8), size=(T, n)),
            "term_obs": rng.normal(size=(in_dim, T, n)).astype(np.float32),
            "action": rng.integers(0, 2, size=(T, n)).astype(np.uint8),
            "reward": rng.normal(size=(T, n)).astype(np.float32)}
    traj = ra.Trajectory(engine, n, T, in_dim)
    traj.write_all(want)
    B = n * T
    x = want["obs"][:, :T, :].reshape(in_dim, B).T
    cnet, pnet = unflatten(cri.get_params(), in_dim, hidden, 1), unflatten(pol.get_params(), in_dim, hidden, 2)
    ra.gae(traj, cri, 0.99, 0.95)
    v = forward64(cnet, x)[0][:, 0]
    assert np.allclose(traj.read(ra.TRAJ_VALUES)[:T].reshape(-1), v, rtol=2e-5, atol=2e-6)
    adv, rtg = traj.read(ra.TRAJ_ADVANTAGES).reshape(-1), traj.read(ra.TRAJ_RETURNS).reshape(-1)
    z, acts = forward64(pnet, x)
    lp = z - np.log(np.exp(z - z.max(axis=1, keepdims=True)).sum(axis=1, keepdims=True)) - z.max(axis=1, keepdims=True)
    p = np.exp(lp)
    onehot = np.eye(2)[want["action"].reshape(-1).astype(np.int64)]
    gwant = backward64(pnet, x, acts, -(adv[:, None].astype(np.float64)) * (onehot - p) / B)
    got = ra.policy_gradient(pol, traj)[0]
    assert got.shape == gwant.shape and np.abs(got - gwant).max() <= 2e-5 * np.abs(gwant).max() + 1e-9
    _, cacts = forward64(cnet, x)
    cwant = backward64(cnet, x, cacts, (2.0 * (v - rtg.astype(np.float64)) / B)[:, None])
    assert np.abs(ra.critic_gradient(cri, traj)[0] - cwant).max() <= 2e-5 * np.abs(cwant).max() + 1e-9
    vec = rng.normal(size=pol.P).astype(np.float32)
    _, tz = jvp64(pnet, unflatten(vec, in_dim, hidden, 2), x)
    fwant = backward64(pnet, x, acts, p * (tz - (p * tz).sum(axis=1, keepdims=True)) / B) + 1e-5 * vec.astype(np.float64)
    assert np.abs(ra.policy_fvp(pol, traj, vec, 1e-5) - fwant).max() <= 5e-5 * np.abs(fwant).max() + 1e-9
    st, losses = ra.critic_update(cri, ra.Adam(cri), traj, 3, want_losses=True)
    assert losses[-1] < losses[0]


def collect(engine, pol, n=96, T=12, seed=3):
    env = ra.CartPoleEnv(engine, n, max_steps=9, seed_env=seed, seed_actor=seed + 1)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    return env, traj


@pytest.mark.parametrize("hidden", [[64, 64], [130], []])
def test_rollout_step_by_step(engine, hidden):
    """the env side of the recorded trajectory replays bit for bit through the oracle's lanes; every recorded action is
    the inverse-CDF draw of the policy's distribution with the lane's actor word (compared wherever the uniform is not
    within 1e-5 of the boundary: the logits are checked to f32 tolerance, not bit for bit)"""
    n, T = 96, 12
    pol = make(engine, 5, hidden, 2, 11)
    env, traj = collect(engine, pol, n, T)
    got = traj.read_all()
    sim = O.LaneSim(n, max_steps=9, seed_env=3, seed_actor=4)
    assert np.array_equal(got["obs"][:, 0, :], sim.observe())
    for t in range(T):
        reward, flag, obs, term = sim.step(got["action"][t])
        assert np.array_equal(got["reward"][t], reward) and np.array_equal(got["flag"][t], flag), t
        assert np.array_equal(got["obs"][:, t + 1, :], obs), t
        m = flag == O.INTERRUPT
        assert np.array_equal(got["term_obs"][:, t, m], term[:, m])
    assert (got["flag"] != O.CONTINUE).any()
    net = unflatten(pol.get_params(), 5, hidden, 2)
    r = O.Prng()
    checked = 0
    for t in range(T):
        z, _ = forward64(net, got["obs"][:, t, :].T)
        p0 = 1.0 / (1.0 + np.exp(z[:, 1] - z[:, 0]))
        for i in range(n):
            O.lib().oracle_prng_seed_from_u64(C.byref(r), 4)
            O.lib().oracle_prng_set_stream(C.byref(r), i)
            O.lib().oracle_prng_set_word_pos(C.byref(r), t)
            u = O.lib().oracle_prng_gen_f32(C.byref(r))
            if abs(u - p0[i]) > 1e-5:
                assert got["action"][t, i] == (0 if u < p0[i] else 1), (t, i)
                checked += 1
    assert checked > 0.99 * n * T
    # a second rollout continues the lanes and the streams
    ra.rollout(env, pol, traj)
    assert np.array_equal(traj.read_all()["obs"][:, 0, :], sim.observe())


def test_rollout_on_the_index_env_lanes(engine):
    """the step-by-step rollout over the Chain lanes' standalone kernels (slip draws from the lane's env stream by global
    step): replayed bit for bit through the oracle's lanes, over two trajectories"""
    n, T = 64, 25
    pol = make(engine, 5, [32, 32], 2, 13)
    env = ra.ChainEnv(engine, n, max_steps=9, seed_env=5, seed_actor=6)
    sim = O.ChainLaneSim(n, max_steps=9, seed_env=5, seed_actor=6)
    traj = ra.Trajectory(engine, n, T, 5)
    for _ in range(2):
        ra.rollout(env, pol, traj)
        got = traj.read_all()
        assert np.array_equal(got["obs"][:, 0, :], sim.observe())
        for t in range(T):
            reward, flag, obs, term = sim.step(got["action"][t])
            assert np.array_equal(got["reward"][t], reward) and np.array_equal(got["flag"][t], flag), t
            assert np.array_equal(got["obs"][:, t + 1, :], obs), t
            m = flag == O.INTERRUPT
            assert np.array_equal(got["term_obs"][:, t, m], term[:, m])
        assert (got["flag"] == O.INTERRUPT).any() and (got["action"] == 1).any() and (got["action"] == 0).any()


# MlpConfig.activation / output_activation (ff/mlp.rs:18-21, ff/activation.rs:11-20): the reference's defaults on every
# shape, then every other variant on the hidden layers and on the output
ACT_CASES = [(h, "Relu", "Identity") for h in ([64, 64], [256], [32, 16, 8], [])] + [
    ([64, 64], "Tanh", "Identity"), ([32, 16, 8], "Sigmoid", "Identity"), ([48], "Identity", "Identity"),
    ([64, 64], "Relu", "Tanh"), ([40, 24], "Tanh", "Sigmoid"), ([128], "Tanh", "Identity"), ([], "Relu", "Tanh")]


@pytest.mark.parametrize("hidden,act,out_act", ACT_CASES)
def test_gradients_and_fisher_vector_products(engine, hidden, act, out_act):
    pol, cri = make(engine, 5, hidden, 2, 21, act, out_act), make(engine, 5, hidden, 1, 22, act, out_act)
    _, traj = collect(engine, pol)
    ra.gae(traj, cri, 0.99, 0.95)
    tr = traj.read_all()
    n, T = tr["action"].shape[1], tr["action"].shape[0]
    B = n * T
    x = tr["obs"][:, :T, :].reshape(5, B).T
    # ---- values, advantages, returns: the oracle's array-fed scan on f64 values
    cnet = unflatten(cri.get_params(), 5, hidden, 1)
    v = forward64(cnet, x, act, out_act)[0][:, 0]
    assert np.allclose(traj.read(ra.TRAJ_VALUES)[:T].reshape(-1), v, rtol=2e-5, atol=2e-6)
    adv, rtg = traj.read(ra.TRAJ_ADVANTAGES).reshape(-1), traj.read(ra.TRAJ_RETURNS).reshape(-1)
    assert np.isfinite(adv).all() and np.abs(adv).max() > 0
    # ---- policy gradient: loss = -mean(A log pi(a)) at ratio 1 (Trpo / Reinforce closure)
    pnet = unflatten(pol.get_params(), 5, hidden, 2)
    z, acts = forward64(pnet, x, act, out_act)
    lp = z - np.log(np.exp(z - z.max(axis=1, keepdims=True)).sum(axis=1, keepdims=True)) - z.max(axis=1, keepdims=True)
    p = np.exp(lp)
    a = tr["action"].reshape(-1).astype(np.int64)
    onehot = np.eye(2)[a]
    dz = -(adv[:, None].astype(np.float64)) * (onehot - p) / B
    want = backward64(pnet, x, acts, dz)
    got, loss = ra.policy_gradient(pol, traj)[:2]
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * scale + 1e-9
    # ---- critic gradient: mean((V - returns)^2)
    dv = (2.0 * (v - rtg.astype(np.float64)) / B)[:, None]
    _, cacts = forward64(cnet, x, act, out_act)
    cwant = backward64(cnet, x, cacts, dv)
    cgot = ra.critic_gradient(cri, traj)[0]
    assert np.abs(cgot - cwant).max() <= 2e-5 * np.abs(cwant).max() + 1e-9
    # ---- Fisher-vector product: J^T (diag(p) - p p^T) J v / B + reg v
    rng = np.random.default_rng(5)
    vec = rng.normal(size=pol.P).astype(np.float32)
    tnet = unflatten(vec, 5, hidden, 2)
    _, tz = jvp64(pnet, tnet, x, act, out_act)
    mz = p * (tz - (p * tz).sum(axis=1, keepdims=True)) / B
    fwant = backward64(pnet, x, acts, mz) + 1e-5 * vec.astype(np.float64)
    fgot = ra.policy_fvp(pol, traj, vec, 1e-5)
    assert np.abs(fgot - fwant).max() <= 5e-5 * np.abs(fwant).max() + 1e-9


@pytest.mark.parametrize("hidden,act,out_act,n,T", [
    ([64, 64], "Relu", "Identity", 3, 5), ([64, 64], "Tanh", "Identity", 700, 9), ([33], "Relu", "Identity", 65, 7),
    ([32, 16, 8], "Sigmoid", "Tanh", 130, 20), ([17, 64], "Relu", "Sigmoid", 96, 12), ([64, 1, 64], "Tanh", "Identity", 50, 3),
    ([100], "Tanh", "Identity", 70, 9), ([128], "Sigmoid", "Tanh", 33, 4)])
def test_fused_matrix_passes_agree_with_the_layer_kernels(engine, hidden, act, out_act, n, T):
    """Shapes the fused matrix-pipe launch takes (kernels_gen_mfma.hip: 1-3 hidden layers of at most 64 units) against
    the per-layer f32 kernels (kernel variant 1; one hidden layer of up to 128 units with other activations than the
    fused module's included) on the same trajectory: two implementations that share no code beyond
    the activation definitions.  Sample counts below one tile, ragged last tiles, fewer tiles than waves; every pass:
    gradient, loss / KL, Fisher-vector product, PPO steps, critic gradient and critic steps."""
    pol, cri = make(engine, 5, hidden, 2, 41, act, out_act), make(engine, 5, hidden, 1, 42, act, out_act)
    env = ra.CartPoleEnv(engine, n, max_steps=9, seed_env=5, seed_actor=6)
    traj = ra.Trajectory(engine, n, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    p0, c0 = pol.get_params(), cri.get_params()
    vec = np.random.default_rng(9).normal(size=pol.P).astype(np.float32)
    moved = (p0 + 0.01 * vec).astype(np.float32)
    got = {}
    for variant in (0, 1):
        engine.set_kernel_variant(variant)
        try:
            pol.set_params(p0)
            cri.set_params(c0)
            r = {"grad": ra.policy_gradient(pol, traj), "fvp": ra.policy_fvp(pol, traj, vec, 0.0),
                 "cgrad": ra.critic_gradient(cri, traj)}
            pol.set_params(moved)
            r["loss_kl"] = ra.policy_loss_kl(pol, traj, p0)
            pol.set_params(p0)
            cfg = ra.ppo_config_default()
            cfg.opt_steps_per_update = 3
            r["ppo"] = ra.ppo_update(pol, ra.Adam(pol), traj, cfg, want_losses=True)[1]
            r["ppo_params"] = pol.get_params()
            r["critic"] = ra.critic_update(cri, ra.Adam(cri), traj, 3, want_losses=True)[1]
            r["critic_params"] = cri.get_params()
            got[variant] = r
        finally:
            engine.set_kernel_variant(0)
    a, b = got[0], got[1]

    def close(x, y, rel):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        return np.abs(x - y).max() <= rel * max(np.abs(y).max(), 1e-30) + 1e-12

    assert close(a["grad"][0], b["grad"][0], 2e-5) and close(a["grad"][1:], b["grad"][1:], 2e-5)  # gradient; loss, entropy
    assert close(a["fvp"], b["fvp"], 5e-5)
    assert close(a["cgrad"][0], b["cgrad"][0], 2e-5) and close(a["cgrad"][1], b["cgrad"][1], 2e-6)
    assert abs(a["loss_kl"][0] - b["loss_kl"][0]) <= 2e-5 * abs(b["loss_kl"][0]) + 1e-7
    assert abs(a["loss_kl"][1] - b["loss_kl"][1]) <= 2e-4 * abs(b["loss_kl"][1]) + 1e-8
    assert close(a["ppo"], b["ppo"], 2e-5) and close(a["critic"], b["critic"], 2e-5)
    # three Adam steps at lr 1e-3: the first step moves every parameter by ~lr whatever the gradient's size, so a
    # gradient entry near zero can flip its step's sign between two f32 summation orders
    assert np.abs(a["ppo_params"] - b["ppo_params"]).max() < 4e-3 and np.abs(a["critic_params"] - b["critic_params"]).max() < 4e-3
    assert np.median(np.abs(a["critic_params"] - b["critic_params"])) < 1e-5


@pytest.mark.parametrize("hidden", [[64, 64], [130]])
def test_updates_run_and_improve(engine, hidden):
    pol, cri = make(engine, 5, hidden, 2, 31), make(engine, 5, hidden, 1, 32)
    env = ra.CartPoleEnv(engine, 256, max_steps=500, seed_env=1, seed_actor=2)
    traj = ra.Trajectory(engine, 256, 32, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    st = ra.trpo_update(pol, traj)
    assert st.status == ra.OPT_OK and 0 < st.constraint_val_final <= 0.01 and st.loss_final < st.loss_initial
    cs, losses = ra.critic_update(cri, ra.Adam(cri), traj, 8, want_losses=True)
    assert losses[-1] < losses[0]
    vcfg = ra.values_opt_config_default()
    vcfg.opt_steps_per_update, vcfg.target, vcfg.discount_factor = 3, ra.VALUE_TARGET_ONE_STEP_TD, 0.99
    vs, vl = ra.values_opt_update(cri, ra.Adam(cri), traj, vcfg, want_losses=True)
    assert np.isfinite(vl).all()
    ppo = ra.ppo_config_default()
    ppo.opt_steps_per_update = 3
    ps, pl = ra.ppo_update(pol, ra.Adam(pol), traj, ppo, want_losses=True)
    assert pl[-1] < pl[0]
    ra.reinforce_update(pol, ra.Adam(pol), traj)
    # actor serialisation round trip (Mlp { layers: [...] })
    doc = ra.actor_to_cbor(env, pol)
    twin = ra.Mlp(engine, 5, hidden, 2)
    ra.module_from_cbor(twin, doc)
    assert np.array_equal(twin.get_params(), pol.get_params())


def test_two_hidden_layers_learn_cartpole(engine):
    """the behavioural check of tests/test_gpu_learning.py on MlpConfig { hidden_sizes: [64, 64] }"""
    n, T = 1024, 64
    env = ra.CartPoleEnv(engine, n, max_steps=500, seed_env=0, seed_actor=1)
    pol, cri = make(engine, 5, [64, 64], 2, 2), make(engine, 5, [64, 64], 1, 3)
    copt = ra.Adam(cri)
    traj = ra.Trajectory(engine, n, T, 5)
    lengths = []
    for _ in range(14):
        ra.rollout(env, pol, traj)
        ends = (traj.read(ra.TRAJ_FLAG) != 0).sum()
        lengths.append(n * T / max(ends, 1))
        ra.gae(traj, cri, 0.99, 0.95)
        ra.trpo_update(pol, traj)
        ra.critic_update(cri, copt, traj, 20)
    assert lengths[0] < 40 and lengths[-1] > 2.5 * lengths[0], lengths


def test_recurrent_policy_with_a_general_critic_on_one_trajectory(engine):
    """the two paths share the trajectory's value planes: a GRU policy and an MlpConfig { hidden_sizes: [32, 32] }
    critic take turns on the same handle"""
    n, T = 64, 20
    env = ra.ChainEnv(engine, n, max_steps=9, seed_env=1, seed_actor=2)
    pol, cri = ra.GruMlp(engine, 5, 2), make(engine, 5, [32, 32], 1, 5)
    pol.init(4)
    traj = ra.Trajectory(engine, n, T, 5)
    for _ in range(2):
        ra.rollout(env, pol, traj)
        ra.gae(traj, cri, 0.95, 0.9)          # general critic first: allocates the shared planes
        cfg = ra.ppo_config_default()
        cfg.opt_steps_per_update = 2
        st, losses = ra.ppo_update(pol, ra.Adam(pol), traj, cfg, want_losses=True)
        assert np.isfinite(losses).all()
        cs, cl = ra.critic_update(cri, ra.Adam(cri), traj, 3, want_losses=True)
        assert cl[-1] < cl[0]


def test_fused_modules_are_untouched_by_a_general_module_on_the_same_trajectory(engine):
    """a general policy regrows the trajectory's update workspace; the fused critic update that follows on the same
    handle gives bit for bit what it gives on a handle no general module has touched"""
    n, T = 256, 32
    results = []
    for with_general in (False, True):
        env = ra.CartPoleEnv(engine, n, max_steps=50, seed_env=8, seed_actor=9)
        beh, cri = make(engine, 5, 128, 2, 1), make(engine, 5, 128, 1, 2)
        traj = ra.Trajectory(engine, n, T, 5)
        ra.rollout(env, beh, traj)
        ra.gae(traj, cri, 0.99, 0.95)
        if with_general:
            big = make(engine, 5, [200, 200], 2, 3)
            st = ra.trpo_update(big, traj)
            assert st.status == ra.OPT_OK
        ra.critic_update(cri, ra.Adam(cri), traj, 5)
        st = ra.trpo_update(beh, traj)
        results.append((cri.get_params(), beh.get_params(), st.step_size))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    assert results[0][2] == results[1][2]


# ---------------------------------------------------------------- LinearConfig { kernel_init, bias_init }
INIT_CASES = [(("Zeros", "FanIn", 0.0), ("Constant", "Constant", 0.25)),
              (("Uniform", "FanIn", 0.0), ("Uniform", "FanOut", 0.0)),
              (("Uniform", "Constant", 0.01), ("Zeros", "FanIn", 0.0)),
              (("Normal", "FanAvg", 0.0), ("Normal", "Constant", 1e-4)),
              (("Orthogonal", "FanIn", 0.0), ("Zeros", "FanIn", 0.0)),
              (("Normal", "FanIn", 0.0), ("Uniform", "FanAvg", 0.0))]


@pytest.mark.parametrize("kinit,binit", INIT_CASES)
def test_initializers_match_the_oracle_and_their_definitions(engine, kinit, binit):
    """rl_mlp_init_with: every `Initializer` variant (src/torch/initializers.rs:8-21,152-176,328-364) with Linear::new's
    fan_in = in + 1 (ff/linear.rs:54-68), bit for bit against oracle_mlp_layers_init, and against what the variants MEAN:
    limits and variances from the shapes, orthonormal rows or columns (compared with numpy's QR of the same normals up
    to the column signs the reference folds away)."""
    for in_dim, hidden, out_dim in ((5, [128], 2), (5, [96, 200], 1), (4, [], 2)):
        m = ra.Mlp(engine, in_dim, hidden if len(hidden) != 1 else hidden[0], out_dim)
        m.init(13, kinit, binit)
        p = m.get_params()
        assert np.array_equal(p, O.mlp_layers_init(in_dim, hidden, out_dim, 13, kinit, binit))
        k = 0
        for fi, fo in layers(in_dim, hidden, out_dim):
            W, b = p[k:k + fi * fo].reshape(fo, fi), p[k + fi * fo:k + fi * fo + fo]
            k += fi * fo + fo
            for (kind, scale, value), t in ((kinit, W), (binit, b)):
                var = {"Constant": value, "FanIn": 1.0 / (fi + 1), "FanOut": 1.0 / fo, "FanAvg": 2.0 / (fi + 1 + fo)}[scale]
                if kind == "Zeros":
                    assert not t.any()
                elif kind == "Constant":
                    assert (t == np.float32(value)).all()
                elif kind == "Uniform":
                    lim = np.float32(np.sqrt(3.0 * var))
                    assert np.abs(t).max() <= lim and (t.size < 64 or np.abs(t).max() > 0.8 * lim)
                    assert t.size < 2000 or abs(t.astype(np.float64).var() / var - 1.0) < 0.15
                elif kind == "Normal":
                    assert t.size < 2000 or abs(t.astype(np.float64).var() / var - 1.0) < 0.15
                    assert t.size < 64 or np.abs(t).max() > 1.5 * np.sqrt(var)  # not a bounded distribution
                else:
                    Wd = W.astype(np.float64)
                    gram = Wd @ Wd.T if fo <= fi else Wd.T @ Wd
                    assert np.abs(gram - np.eye(gram.shape[0])).max() < 1e-5


def test_initializer_argument_checks(engine):
    m = ra.Mlp(engine, 5, 128, 2)
    k = ra.Initializer.of(("Uniform", "FanAvg", 0.0))
    # bias_init: None belongs to modules built without bias vectors (rl_mlp_create_config: test_layers_without_bias_vectors)
    assert ra.lib().rl_mlp_init_with(m.h, C.c_uint64(1), C.byref(k), None) == ra.ERR_INVALID_ARGUMENT
    with pytest.raises(ra.RelearnError):  # init_orthogonal needs two dimensions (initializers.rs:331-334)
        m.init(1, ("Uniform", "FanAvg", 0.0), ("Orthogonal", "FanIn", 0.0))
    bad = ra.Initializer(7, 0, 0.0)
    assert ra.lib().rl_mlp_init_with(m.h, C.c_uint64(1), C.byref(bad), C.byref(k)) == ra.ERR_INVALID_ARGUMENT
    g = ra.GruMlp(engine, 5, 2)  # the recurrent chains keep RnnBaseConfig::default's initializers
    assert ra.lib().rl_mlp_init_with(g.h, C.c_uint64(1), C.byref(k), C.byref(k)) == ra.ERR_UNSUPPORTED
    # the default pair is rl_mlp_init
    a, b = ra.Mlp(engine, 5, 128, 2), ra.Mlp(engine, 5, 128, 2)
    a.init(5)
    b.init(5, ("Uniform", "FanAvg", 0.0), ("Uniform", "FanAvg", 0.0))
    assert np.array_equal(a.get_params(), b.get_params())


# ------------------------------------------------------------------ LinearConfig::bias_init = None (ff/linear.rs:13-33)
def with_zero_biases(p, in_dim, hidden, out_dim):
    """the bias-less flat vector with a zero bias vector inserted after every kernel: what a module WITH biases holds when
    they are all zero — `acc = 0; acc = fma(x_k, w_k, acc)` is the bias-less chain"""
    out, k = [], 0
    for fi, fo in layers(in_dim, hidden, out_dim):
        out += [p[k:k + fi * fo], np.zeros(fo, dtype=np.float32)]
        k += fi * fo
    assert k == p.size
    return np.concatenate(out)


def drop_bias_entries(g, in_dim, hidden, out_dim):
    out, k = [], 0
    for fi, fo in layers(in_dim, hidden, out_dim):
        out.append(g[k:k + fi * fo])
        k += fi * fo + fo
    return np.concatenate(out)


@pytest.mark.parametrize("hidden,act,out_act", [([64, 64], "Relu", "Identity"), ([100], "Tanh", "Identity"),
                                                ([32, 16, 8], "Sigmoid", "Tanh"), ([], "Relu", "Identity")])
def test_layers_without_bias_vectors(engine, hidden, act, out_act):
    """MlpConfig with LinearConfig { bias_init: None }: the flat vector holds the kernels only (Linear::
    trainable_variables, linear.rs:104-117); forward bit-identical to the oracle's network with all-zero biases;
    gradients, Fisher-vector products and the critic gradient against the f64 NumPy network, the bias entries dropped;
    the actor document carries `bias: null` and loads back; updates move the kernels."""
    from cbor_ref import decode
    pol = ra.Mlp(engine, 5, hidden, 2, act, out_act, bias=False)
    cri = ra.Mlp(engine, 5, hidden, 1, act, out_act, bias=False)
    pol.init(21)
    cri.init(22, kernel_init=("Normal", "FanIn", 0.0))
    assert pol.P == sum(fi * fo for fi, fo in layers(5, hidden, 2)) and cri.P == sum(fi * fo for fi, fo in layers(5, hidden, 1))
    p, c = pol.get_params(), cri.get_params()
    k = 0
    for fi, fo in layers(5, hidden, 2):  # Linear::new: Uniform(+-sqrt(6 / (fan_in + 1 + fan_out))), kernels only
        lim = np.float32(np.sqrt(3.0 * (2.0 / ((fi + 1) + fo))))
        assert np.abs(p[k:k + fi * fo]).max() <= lim
        k += fi * fo
    with pytest.raises(ra.RelearnError):  # a bias initializer for a module that has no bias vectors
        ra._check(ra.lib().rl_mlp_init_with(pol.h, C.c_uint64(1), C.byref(ra.Initializer.of(("Zeros", "Constant", 0.0))),
                                            C.byref(ra.Initializer.of(("Zeros", "Constant", 0.0)))), engine.h)
    x = (2.0 * np.random.default_rng(4).normal(size=(257, 5))).astype(np.float32)
    want = O.mlp_layers_forward(5, hidden, 2, with_zero_biases(p, 5, hidden, 2), x, act, out_act)
    assert np.array_equal(pol.forward(x), want)
    env, traj = collect(engine, pol)
    ra.gae(traj, cri, 0.99, 0.95)
    tr = traj.read_all()
    n, T = tr["action"].shape[1], tr["action"].shape[0]
    B = n * T
    xs = tr["obs"][:, :T, :].reshape(5, B).T
    adv, rtg = traj.read(ra.TRAJ_ADVANTAGES).reshape(-1), traj.read(ra.TRAJ_RETURNS).reshape(-1)
    pnet = unflatten(with_zero_biases(p, 5, hidden, 2), 5, hidden, 2)
    cnet = unflatten(with_zero_biases(c, 5, hidden, 1), 5, hidden, 1)
    z, acts = forward64(pnet, xs, act, out_act)
    zmax = z.max(axis=1, keepdims=True)
    lp = z - zmax - np.log(np.exp(z - zmax).sum(axis=1, keepdims=True))
    pr = np.exp(lp)
    a = tr["action"].reshape(-1).astype(np.int64)
    want_g = drop_bias_entries(backward64(pnet, xs, acts, -(adv[:, None].astype(np.float64)) * (np.eye(2)[a] - pr) / B), 5, hidden, 2)
    got_g = ra.policy_gradient(pol, traj)[0]
    assert got_g.shape == want_g.shape and np.abs(got_g - want_g).max() <= 2e-5 * np.abs(want_g).max() + 1e-9
    vec = np.random.default_rng(5).normal(size=pol.P).astype(np.float32)
    tnet = unflatten(with_zero_biases(vec, 5, hidden, 2), 5, hidden, 2)
    _, tz = jvp64(pnet, tnet, xs, act, out_act)
    want_f = drop_bias_entries(backward64(pnet, xs, acts, pr * (tz - (pr * tz).sum(axis=1, keepdims=True)) / B), 5, hidden, 2)
    got_f = ra.policy_fvp(pol, traj, vec, 0.0)
    assert np.abs(got_f - want_f).max() <= 5e-5 * np.abs(want_f).max() + 1e-9
    v, cacts = forward64(cnet, xs, act, out_act)
    want_c = drop_bias_entries(backward64(cnet, xs, cacts, 2.0 * (v - rtg[:, None].astype(np.float64)) / B), 5, hidden, 1)
    got_c = ra.critic_gradient(cri, traj)[0]
    assert np.abs(got_c - want_c).max() <= 2e-5 * np.abs(want_c).max() + 1e-9
    # the document: Linear { kernel, bias: None }
    doc = ra.actor_to_cbor(env, pol)

    def find(vv):
        if isinstance(vv, dict):
            if "activation" in vv and "layers" in vv:
                return vv
            vv = list(vv.values())
        if isinstance(vv, (list, tuple)):
            for item in vv:
                hit = find(item)
                if hit is not None:
                    return hit
        return None
    mod = find(decode(bytes(doc)))
    assert mod is not None and all(layer["bias"] is None for layer in mod["layers"])
    twin = ra.Mlp(engine, 5, hidden, 2, act, out_act, bias=False)
    ra.module_from_cbor(twin, doc)
    assert np.array_equal(twin.get_params(), p)
    biased = ra.Mlp(engine, 5, hidden if hidden else [7], 2, act, out_act)
    if hidden:
        with pytest.raises(ra.RelearnError):  # a document without bias vectors is not a module's that has them
            ra.module_from_cbor(biased, doc)
    # updates work on the kernels-only vector
    st = ra.trpo_update(pol, traj)
    cs, losses = ra.critic_update(cri, ra.Adam(cri), traj, 3, want_losses=True)
    assert st.status in (ra.OPT_OK, ra.OPT_LOSS_NOT_IMPROVING, ra.OPT_CONSTRAINT_VIOLATED)
    assert losses[-1] < losses[0] and not np.array_equal(cri.get_params(), c)
