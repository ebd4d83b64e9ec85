"""The numeric range of the fused update kernels (include/relearn_hip.h, "Numeric range of the fused kernels").

The fused critic / policy kernels run layer 1 on weights scaled by 2^96 and take relu' from a clamped conversion
(relearn_amd/csrc/bf16_tile.hpp).  Plain f32 — the reference's `Mlp::forward`, src/torch/modules/ff/mlp.rs:139-151 — has
no such scaling, so the library states the range in which the scaled form is exact, checks it per launch from the weights
and the magnitudes of the trajectory's observations (range_guard), and refuses (RL_ERR_UNSUPPORTED) outside it instead of
returning a wrong mask silently.  Here: host-fed histories (rl_traj_write) with |obs| up to 1e6 and down to 1e-30 and
weights up to 1e3 — inside the range the critic gradient, the policy gradient and the Fisher-vector product agree with
the f64 oracle like they do at ordinary magnitudes; outside it every entry point reports the error, kernel variant 1
computes the same inputs, and the engine carries on afterwards."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

H, N, T = 128, 1024, 32
PS, CS = O.MlpShape(5, H, 2), O.MlpShape(5, H, 1)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def history(feature_scales, seed=0, nan_at=None):
    """a synthetic history: observations N(0, 1) x a scale per feature, random actions, advantages and returns"""
    rng = np.random.default_rng(seed)
    obs = (rng.standard_normal((5, T + 1, N)) * np.asarray(feature_scales, dtype=np.float64)[:, None, None]).astype(np.float32)
    if nan_at is not None:
        obs[nan_at] = np.nan
    return dict(obs=obs, action=rng.integers(0, 2, size=(T, N)).astype(np.uint8),
                reward=np.ones((T, N), dtype=np.float32), flag=np.zeros((T, N), dtype=np.uint8),
                term_obs=np.zeros((5, T, N), dtype=np.float32),
                adv=rng.standard_normal((T, N)).astype(np.float32), rtg=(10.0 * rng.standard_normal((T, N))).astype(np.float32))


def modules(engine, w1_scale=1.0, w2_scale=1.0, zero_bias=False):
    pol, cri = ra.Mlp(engine, 5, H, 2), ra.Mlp(engine, 5, H, 1)
    pol.init(2)
    cri.init(3)
    for m in (pol, cri):
        p = m.get_params()
        p[:5 * H] *= np.float32(w1_scale)
        if zero_bias:
            p[5 * H:6 * H] = 0.0
        p[6 * H:] *= np.float32(w2_scale)
        m.set_params(p)
    return pol, cri


def load(engine, h):
    traj = ra.Trajectory(engine, N, T, 5)
    traj.write_all(h)
    traj.write(ra.TRAJ_ADVANTAGES, h["adv"])
    traj.write(ra.TRAJ_RETURNS, h["rtg"])
    return traj


def flat(h):
    x, a = O.flat_samples(h)
    return x, a.astype(np.uint8), h["adv"].reshape(-1), h["rtg"].reshape(-1)


IN_RANGE = {
    # |obs| up to ~5e6 on every feature, Glorot weights: sum |w| max|x| ~ 5e6 << 2^31
    # (the output layer is scaled down with the inputs' scale: logits of order 1e6 saturate the softmax, the Fisher-vector
    # product is then carried by the one or two samples with nearly tied logits, and two correct f32 evaluations of it
    # differ in the fourth digit — a property of the problem, not of the kernels)
    "obs_1e6": dict(scales=[1e6] * 5, w2_scale=1e-6),
    # |obs| down to 1e-30: every product w x is far below the bias, which carries the pre-activation (non-zero Glorot bias)
    "obs_1e-30": dict(scales=[1e-30] * 5),
    # each feature on its own scale
    "obs_mixed": dict(scales=[1e6, 1.0, 1e-30, 1e-3, 1e3], w2_scale=1e-6),
    # layer-1 weights up to ~2e2 ... 1e3 (x 1000 on Glorot's 0.2), the output layer scaled down to keep the logits finite
    "w1_1e3": dict(scales=[1.0] * 5, w1_scale=5e3, w2_scale=1e-4),
    # both at once, still inside: 5 x 20 x 5e6 = 5e8 < 2^31
    "obs_1e6_w1_2e1": dict(scales=[1e6] * 5, w1_scale=1e2, w2_scale=1e-7),
}


@pytest.mark.parametrize("case", sorted(IN_RANGE))
def test_fused_kernels_inside_their_range_match_the_f64_oracle(engine, case):
    c = IN_RANGE[case]
    h = history(c["scales"])
    pol, cri = modules(engine, c.get("w1_scale", 1.0), c.get("w2_scale", 1.0))
    traj = load(engine, h)
    x, a, adv, rtg = flat(h)
    pp, cp = pol.get_params(), cri.get_params()
    assert np.abs(pp[:5 * H]).max() <= 1.1e3
    # the yardstick of tests/test_gpu_parity.py: no farther from the f64 truth than 4 x what a correct f32 evaluation is
    g_d, loss_d, _ = ra.policy_gradient(pol, traj)
    g64, l64 = O.grad_f64_mt("policy", PS, pp, x, a, adv)
    g32, _ = O.grad_f64_mt("policy", PS, pp, x, a, adv, f32_samples=True)
    assert np.all(np.isfinite(g_d)) and rel_err(g_d, g64) < max(4 * rel_err(g32, g64), 1e-6), case
    assert abs(loss_d - l64) <= 1e-5 * max(1.0, abs(l64))
    v = np.random.default_rng(7).standard_normal(len(pp)).astype(np.float32)
    h_d = ra.policy_fvp(pol, traj, v, 0.0)
    h64, _ = O.grad_f64_mt("fvp", PS, pp, x, v=v)
    h32, _ = O.grad_f64_mt("fvp", PS, pp, x, v=v, f32_samples=True)
    assert np.all(np.isfinite(h_d)) and rel_err(h_d, h64) < max(4 * rel_err(h32, h64), 1e-6), case
    gc_d, lc_d = ra.critic_gradient(cri, traj)
    gc64, lc64 = O.grad_f64_mt("critic", CS, cp, x, aux=rtg)
    gc32, _ = O.grad_f64_mt("critic", CS, cp, x, aux=rtg, f32_samples=True)
    assert np.all(np.isfinite(gc_d)) and rel_err(gc_d, gc64) < max(4 * rel_err(gc32, gc64), 1e-6), case
    assert abs(lc_d - lc64) <= 1e-5 * lc64
    # the fused kernels really ran (variant 1 gives other low-order bits on the same input)
    engine.set_kernel_variant(1)
    try:
        gc_v1, _ = ra.critic_gradient(cri, traj)
    finally:
        engine.set_kernel_variant(0)
    assert rel_err(gc_v1, gc64) < max(4 * rel_err(gc32, gc64), 1e-6)


OUT_OF_RANGE = {
    # 5 x ~200 x 5e6 = 5e9 >= 2^31: the scaled accumulator could overflow
    "overflow": dict(scales=[1e6] * 5, w1_scale=1e3, w2_scale=1e-9),
    # no bias and |w x| ~ 1e-31: a non-zero pre-activation below 2^-96 would give a fractional relu' mask
    "tiny_pre": dict(scales=[1e-30] * 5, zero_bias=True),
    # a NaN observation
    "nan_obs": dict(scales=[1.0] * 5, nan_at=(2, 3, 5)),
}


@pytest.mark.parametrize("case", sorted(OUT_OF_RANGE))
def test_outside_the_range_is_an_error_never_a_silent_mask(engine, case):
    c = OUT_OF_RANGE[case]
    h = history(c["scales"], nan_at=c.get("nan_at"))
    pol, cri = modules(engine, c.get("w1_scale", 1.0), c.get("w2_scale", 1.0), c.get("zero_bias", False))
    traj = load(engine, h)
    v = np.ones(pol.P, dtype=np.float32)
    for call in (lambda: ra.critic_gradient(cri, traj), lambda: ra.policy_gradient(pol, traj),
                 lambda: ra.policy_fvp(pol, traj, v, 0.0), lambda: ra.trpo_update(pol, traj),
                 lambda: ra.critic_update(cri, ra.Adam(cri), traj, 2)):
        with pytest.raises(ra.RelearnError) as err:
            call()
        assert err.value.code == ra.ERR_UNSUPPORTED and "numeric range" in str(err.value), case
    # plain f32 arithmetic has no such bound: the v1 kernels take the same inputs
    if case != "nan_obs":
        x, a, adv, rtg = flat(h)
        engine.set_kernel_variant(1)
        try:
            gc_v1, _ = ra.critic_gradient(cri, traj)
            g_v1, _, _ = ra.policy_gradient(pol, traj)
        finally:
            engine.set_kernel_variant(0)
        gc64, _ = O.grad_f64_mt("critic", CS, cri.get_params(), x, aux=rtg)
        gc32, _ = O.grad_f64_mt("critic", CS, cri.get_params(), x, aux=rtg, f32_samples=True)
        assert np.all(np.isfinite(gc_v1)) and np.all(np.isfinite(g_v1))
        assert rel_err(gc_v1, gc64) < max(4 * rel_err(gc32, gc64), 1e-5)
    # the error is not sticky: the engine goes on with a history inside the range
    ok = load(engine, history([1.0] * 5, seed=1))
    pol2, cri2 = modules(engine)
    gc, _ = ra.critic_gradient(cri2, ok)
    g, _, _ = ra.policy_gradient(pol2, ok)
    assert np.all(np.isfinite(gc)) and np.all(np.isfinite(g))


def test_rollout_histories_are_measured_again_every_period(engine):
    """the range words follow the planes: a rollout after a host-fed history of huge observations is judged on its own
    observations, and a host-fed history after a rollout on its own"""
    env = ra.CartPoleEnv(engine, N, max_steps=60, seed_env=3, seed_actor=4)
    pol, cri = modules(engine, w1_scale=1e3, w2_scale=1e-9)  # fine for CartPole's |obs| < 2^4, not for |obs| ~ 1e6
    traj = ra.Trajectory(engine, N, T, 5)
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    ra.critic_gradient(cri, traj)
    traj.write(ra.TRAJ_OBS, history([1e6] * 5)["obs"])
    with pytest.raises(ra.RelearnError) as err:
        ra.critic_gradient(cri, traj)
    assert err.value.code == ra.ERR_UNSUPPORTED
    ra.rollout(env, pol, traj)
    ra.gae(traj, cri, 0.99, 0.95)
    gc, _ = ra.critic_gradient(cri, traj)
    assert np.all(np.isfinite(gc))


def test_dqn_update_checks_its_weights_against_the_measured_observation_range(engine):
    """rl_dqn_update's fused step: the collection kernel folds the magnitude range of every observation it puts into the
    replay store into the workspace's range words (rounds 3-5 kept fixed bounds 2^-64 .. 2^16 there, under which a zero
    bias could never pass: ADVICE round 5), and the guard checks the action-value network's weights against it.  The default
    initialisation passes; a network with a ZERO bias vector — `bias_init: Some(Initializer::Zeros)`, a legal reference
    configuration (src/torch/modules/ff/linear.rs:13-33) — passes and trains like the oracle's f32 network; a network
    whose layer-1 rows could carry a pre-activation past 2^31 is refused with the parameters and the optimiser state put
    back as they were (all or nothing), and variant 1 trains that one."""
    def agent(w1_scale, zero_bias=False):
        env = ra.CartPoleEnv(engine, 256, max_steps=60, seed_env=9, seed_actor=10)
        q = ra.Mlp(engine, 5, H, 2)
        q.init(77)
        p = q.get_params()
        p[:5 * H] *= np.float32(w1_scale)
        p[6 * H:] *= np.float32(1.0 / w1_scale)
        if zero_bias:
            p[5 * H:6 * H] = 0.0
        q.set_params(p)
        cfg = ra.dqn_config_default()
        cfg.exploration_kind, cfg.exploration_start = ra.SCHEDULE_CONSTANT, 0.5
        cfg.minibatch_steps, cfg.opt_steps_per_update, cfg.buffer_capacity = 2000, 3, 128
        dqn = ra.Dqn(env, q, ra.Adam(q), cfg)
        dqn.collect(60)
        return dqn, q

    st, losses = agent(1.0)[0].update(want_losses=True)
    assert st.opt_steps == 3 and np.all(np.isfinite(losses))
    # zero bias: passes the guard (CartPole's observations are nowhere near 2^-46 / max|w|), and the fused step computes
    # what variant 1's plain f32 kernels compute on the same minibatches
    zb, qz = agent(1.0, zero_bias=True)
    st, losses = zb.update(want_losses=True)
    assert st.opt_steps == 3 and np.all(np.isfinite(losses)) and np.mean(qz.get_params()[5 * H:6 * H] != 0.0) > 0.5
    engine.set_kernel_variant(1)
    try:
        zb1, qz1 = agent(1.0, zero_bias=True)
        st1, losses1 = zb1.update(want_losses=True)
    finally:
        engine.set_kernel_variant(0)
    # (Adam turns a rounding-level difference of a near-zero gradient entry into a step of the order of the learning rate:
    # losses to 1e-5, parameters entry by entry for all but a few)
    assert np.allclose(losses, losses1, rtol=1e-5)
    assert np.mean(np.abs(qz.get_params() - qz1.get_params()) < 2e-5) > 0.97
    # rows of ~2e8: 5 x 2e8 x max|obs| (a few units) >= 2^31
    big, qb = agent(1e9)
    before = qb.get_params()
    with pytest.raises(ra.RelearnError) as err:
        big.update()
    assert err.value.code == ra.ERR_UNSUPPORTED and "numeric range" in str(err.value)
    assert np.array_equal(qb.get_params(), before)  # nothing of the refused update stays
    engine.set_kernel_variant(1)
    try:
        st, losses = agent(1e9)[0].update(want_losses=True)
    finally:
        engine.set_kernel_variant(0)
    assert st.opt_steps == 3 and np.all(np.isfinite(losses))


def test_critic_update_without_statistics_still_reports_the_range_error(engine):
    """rl_critic_update / rl_values_opt_update with stats == NULL and losses == NULL used to skip the guard's error word and
    leave the guard disarmed for every later call of that kind: the call itself must return
    RL_ERR_UNSUPPORTED, and the next call on a history inside the range must work"""
    import ctypes as C
    c = OUT_OF_RANGE["overflow"]
    pol, cri = modules(engine, c.get("w1_scale", 1.0), c.get("w2_scale", 1.0))
    traj = load(engine, history(c["scales"]))
    opt = ra.Adam(cri)
    code = ra.lib().rl_critic_update(cri.h, opt.h, traj.h, C.c_uint64(2), None, None)
    assert code == ra.ERR_UNSUPPORTED
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update = 2
    assert ra.lib().rl_values_opt_update(cri.h, opt.h, traj.h, C.byref(ccfg), None, None) == ra.ERR_UNSUPPORTED
    pol2, cri2 = modules(engine)
    ok = load(engine, history([1.0] * 5, seed=1))
    assert ra.lib().rl_critic_update(cri2.h, ra.Adam(cri2).h, ok.h, C.c_uint64(2), None, None) == ra.OK
    # (and the guard is armed again: the out-of-range history is still refused by a call that asks for nothing)
    assert ra.lib().rl_critic_update(cri.h, opt.h, traj.h, C.c_uint64(1), None, None) == ra.ERR_UNSUPPORTED


def scaled(engine, out_dim, seed, w1_scale=1.0, w2_scale=1.0):
    m = ra.Mlp(engine, 5, H, out_dim)
    m.init(seed)
    p = m.get_params()
    p[:5 * H] *= np.float32(w1_scale)
    p[6 * H:] *= np.float32(w2_scale)
    m.set_params(p)
    return m


UPDATES = {
    # (module's output width, the update through the C ABI)
    "critic_update": (1, lambda m, opt, traj: ra.critic_update(m, opt, traj, 3)),
    "ppo_update": (2, lambda m, opt, traj: ra.ppo_update(m, opt, traj)),
    "reinforce_update": (2, lambda m, opt, traj: ra.reinforce_update(m, opt, traj)),
}


@pytest.mark.parametrize("which", sorted(UPDATES))
def test_a_refused_update_is_refused_whole(engine, which):
    """the guard reported after the fact — by the time RL_ERR_UNSUPPORTED came back the module had been
    stepped K times on invalid masks.  Now the launch that finds the violation sets a veto word on the device and the
    optimiser kernels behind it apply nothing: parameters bit for bit as before, and — seen through what follows — the Adam
    moments and the step count too: [2 good updates, a refused one, 2 good updates] leaves exactly the parameters of
    [2 good updates, 2 good updates] on a twin."""
    out_dim, update = UPDATES[which]
    good, bad = history([1.0] * 5, seed=1), history([1e10] * 5, seed=2)  # 5 x 0.2 x 1e10 >= 2^31 for Glorot rows
    tg, tb = load(engine, good), load(engine, bad)
    m, twin = scaled(engine, out_dim, 11), scaled(engine, out_dim, 11)
    opt, opt_twin = ra.Adam(m), ra.Adam(twin)
    for _ in range(2):
        update(m, opt, tg)
        update(twin, opt_twin, tg)
    before = m.get_params()
    assert np.array_equal(before, twin.get_params())
    with pytest.raises(ra.RelearnError) as err:
        update(m, opt, tb)
    assert err.value.code == ra.ERR_UNSUPPORTED and "not applied" in str(err.value)
    assert ("critic chain" if out_dim == 1 else "policy chain") in str(err.value)
    assert np.array_equal(m.get_params(), before)
    for _ in range(2):
        update(m, opt, tg)
        update(twin, opt_twin, tg)
    assert np.array_equal(m.get_params(), twin.get_params())
    assert not np.array_equal(m.get_params(), before)


def test_a_refused_trpo_step_leaves_the_policy_alone(engine):
    pol = scaled(engine, 2, 12)
    tb = load(engine, history([1e10] * 5, seed=2))
    before = pol.get_params()
    with pytest.raises(ra.RelearnError) as err:
        ra.trpo_update(pol, tb)
    assert err.value.code == ra.ERR_UNSUPPORTED and "policy chain" in str(err.value)
    assert np.array_equal(pol.get_params(), before)
    st = ra.trpo_update(pol, load(engine, history([1.0] * 5, seed=1)))
    assert st.status in (ra.OPT_OK, ra.OPT_LOSS_NOT_IMPROVING, ra.OPT_CONSTRAINT_VIOLATED)


@pytest.mark.parametrize("refused", ["critic", "policy"])
def test_the_two_chains_of_the_combined_update_have_a_guard_word_each(engine, refused):
    """rl_actor_critic_update runs the policy chain and the critic chain side by side on two streams.  With one error word
    per trajectory the TRPO chain's read-back could pick up the critic chain's violation; each chain now
    has its own: a module of ONE chain out of range -> the error names that chain, that module is untouched, the other
    chain's step stands, and nothing of it lingers — the same call with both modules in range then works."""
    traj = load(engine, history([1e3] * 5, seed=3))
    # in range at |obs| ~ 5e3: Glorot rows (5 x 0.2 x 5e3 << 2^31; the policy's logits scaled back to order 1); out of it:
    # rows x 1e7 (5 x 2e6 x 5e3 = 5e10 >= 2^31), the output layer scaled down to keep the numbers finite
    bad, fine = dict(w1_scale=1e7, w2_scale=1e-10), dict(w2_scale=1e-3)
    pol = scaled(engine, 2, 13, **(bad if refused == "policy" else fine))
    cri = scaled(engine, 1, 14, **(bad if refused == "critic" else {}))
    opt = ra.Adam(cri)
    ccfg = ra.values_opt_config_default()
    ccfg.opt_steps_per_update = 3
    p0, c0 = pol.get_params(), cri.get_params()
    with pytest.raises(ra.RelearnError) as err:
        ra.actor_critic_update(pol, cri, opt, traj, critic_cfg=ccfg)
    assert err.value.code == ra.ERR_UNSUPPORTED and (refused + " chain") in str(err.value)
    if refused == "critic":
        assert np.array_equal(cri.get_params(), c0)
    else:
        assert np.array_equal(pol.get_params(), p0)
        assert not np.array_equal(cri.get_params(), c0)  # (the critic chain ran beside it, like under a NaN policy step)
    pol2, cri2 = scaled(engine, 2, 13, **fine), scaled(engine, 1, 14)
    out = ra.actor_critic_update(pol2, cri2, ra.Adam(cri2), traj, critic_cfg=ccfg)
    assert out is not None and np.all(np.isfinite(cri2.get_params())) and np.all(np.isfinite(pol2.get_params()))
