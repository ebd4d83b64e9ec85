"""GPU parity tests of the DQN path (BASELINE.json configs[2]): epsilon-greedy collection into the HBM replay
rings, the parallel minibatch sampler, value targets, the fused MSE gradient and the Adam loop — through the C ABI,
against oracle/dqn.c on the same seeds.

Bars: bit-exact for everything integer or on the rollout path (actions, flags, ring bookkeeping, episode picks,
Prng positions, reward-to-go and TD targets — identical op order on both sides); fp32 tolerance, stated below, for
sums over the minibatch (gradient, loss, parameters after Adam).
"""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

ra = pytest.importorskip("relearn_amd")

KEY = [0x9E3779B9, 0x7F4A7C15, 3, 4, 5, 6, 7, 0xFFFFFFFF]
GRAD_RTOL = 2e-6   # vs the f64 evaluation of the same minibatch, relative to max |g|
PARAM_ATOL = 2e-5  # after a handful of Adam steps at lr 1e-3 (sign-sensitive where |g| is tiny)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.fixture(params=[0, 1], ids=["kernels-best", "kernels-v1"])
def variant(engine, request):
    engine.set_kernel_variant(request.param)
    yield request.param
    engine.set_kernel_variant(0)


def make(engine, n=256, hidden=128, capacity=64, minibatch=500, max_steps=23, td=False, opt_steps=4, limit=None,
         eps=("const", 0.3), lane_offset=0, episode_capacity=0):
    limit = ra.LIMIT_VISIBLE if limit is None else limit
    D = 5 if limit == ra.LIMIT_VISIBLE else 4
    env = ra.CartPoleEnv(engine, n, max_steps=max_steps, limit=limit, seed_env=21, seed_actor=34,
                         lane_offset=lane_offset)
    sim = O.LaneSim(n, max_steps=max_steps, limit=limit, seed_env=21, seed_actor=34, lane_offset=lane_offset)
    qs = O.MlpShape(D, hidden, 2)
    q = ra.Mlp(engine, D, hidden, 2)
    q.init(77)
    opt = ra.Adam(q)
    cfg = ra.dqn_config_default()
    cfg.target = ra.DQN_TARGET_ONE_STEP_TD if td else ra.DQN_TARGET_REWARD_TO_GO
    if eps[0] == "const":
        cfg.exploration_kind, cfg.exploration_start = ra.SCHEDULE_CONSTANT, eps[1]
    else:
        cfg.exploration_kind = ra.SCHEDULE_LINEAR_ANNEALED
        cfg.exploration_start, cfg.exploration_end, cfg.exploration_period = eps[1:]
    cfg.minibatch_steps = minibatch
    cfg.opt_steps_per_update = opt_steps
    cfg.buffer_capacity = capacity
    cfg.episode_capacity = episode_capacity
    cfg.discount_factor = 0.99
    for i, k in enumerate(KEY):
        cfg.agent_key[i] = k
    dqn = ra.Dqn(env, q, opt, cfg)
    osim = O.DqnSim(sim, qs, O.mlp_init(qs, 77), capacity, KEY, minibatch, gamma=np.float32(0.99), one_step_td=td)
    return dqn, osim


def check_store(dqn, osim):
    """ring bookkeeping and contents of every lane against the oracle's ReplayBuffers"""
    head, count = dqn.replay_read(ra.REPLAY_HEAD), dqn.replay_read(ra.REPLAY_COUNT)
    eph, epc = dqn.replay_read(ra.REPLAY_EP_HEAD), dqn.replay_read(ra.REPLAY_EP_COUNT)
    total, ep_end = dqn.replay_read(ra.REPLAY_TOTAL), dqn.replay_read(ra.REPLAY_EP_END)
    obs, nobs = dqn.replay_read(ra.REPLAY_OBS), dqn.replay_read(ra.REPLAY_NEXT_OBS)
    act, rew, flag = (dqn.replay_read(f) for f in (ra.REPLAY_ACTION, ra.REPLAY_REWARD, ra.REPLAY_FLAG))
    assert np.array_equal(dqn.replay_read(ra.REPLAY_ACTOR_POS), osim.actor_pos())
    Cc, E = dqn.C, dqn.E
    for i in range(dqn.n):
        ns, ne, tot = osim.lane_info(i)
        assert (count[i], epc[i], total[i]) == (ns, ne, tot)
        tags, lens = osim.lane_dump(i)
        if ns:
            assert head[i] == tags[0]
        ends = [int(ep_end[(eph[i] + k) % E, i]) for k in range(ne)]
        assert ends == list(int(head[i]) + np.cumsum(lens.astype(np.int64)))
        for k in tags[:: max(1, ns // 7)]:  # a spread of the stored steps, compared field by field
            o, a, r, nx, no = osim.step_data(i, int(k))
            slot = int(k) % Cc
            assert np.array_equal(obs[:, slot, i], o) and act[slot, i] == a and rew[slot, i] == r
            assert flag[slot, i] == nx
            if nx == O.INTERRUPT:
                assert np.array_equal(nobs[:, slot, i], no)


@pytest.mark.parametrize("limit", [ra.LIMIT_VISIBLE, ra.LIMIT_NONE])
def test_collection_bit_exact_with_eviction(engine, limit):
    # n is not a multiple of the wave size; capacity 48 forces evictions from the second collection on
    for eps, T in [(1.0, 30), (0.5, 45), (0.0, 31)]:
        dqn, osim = make(engine, n=200, capacity=48, limit=limit, max_steps=23, eps=("const", eps))
        for rep in range(3):
            st = dqn.collect(T)
            flags_o, full = osim.collect(T, eps)
            assert not full and st.exploration_rate == eps
            flags_d = dqn.replay_read(ra.REPLAY_LAST_FLAGS)
            assert np.array_equal(flags_d, flags_o)
            assert st.steps == T * dqn.n and st.episodes_ended == int((flags_o != 0).sum())
            check_store(dqn, osim)
        st_d, st_o = dqn.env.get_state(), osim.sim.get_state()
        for a, b in zip(st_d, st_o):
            assert np.array_equal(a, b)
        dqn.close()


def test_buffer_full_is_an_error(engine):
    dqn, osim = make(engine, n=64, capacity=6, max_steps=500, eps=("const", 0.0))
    with pytest.raises(ra.RelearnError) as e:
        dqn.collect(30)
    assert e.value.code == ra.ERR_BUFFER_FULL
    assert osim.collect(30, 0.0)[1]


@pytest.mark.parametrize("td", [False, True], ids=["reward-to-go", "one-step-td"])
def test_update_on_an_empty_store_fails_before_any_step(engine, td):
    """sampling from lanes without a complete episode (Uniform::new(0, 0) panics in the reference) is reported by the
    first chunk of the pipelined draws: no optimisation step has run, the parameters are untouched, and the agent works
    normally afterwards"""
    dqn, osim = make(engine, n=64, capacity=64, minibatch=300, opt_steps=5, td=td)
    before = dqn.qnet.get_params()
    with pytest.raises(ra.RelearnError) as e:
        dqn.update()
    assert e.value.code == ra.ERR_INVALID_ARGUMENT
    assert np.array_equal(dqn.qnet.get_params(), before)
    dqn.collect(40)
    st = dqn.update()
    assert st.opt_steps == 5 and np.isfinite(st.loss_last)
    assert not np.array_equal(dqn.qnet.get_params(), before)


def test_a_sampler_failure_in_a_later_chunk_leaves_the_agent_as_it_was(engine):
    """With pipelined draws the chunks of an update are validated one by one while earlier chunks already train.  A
    failure in a later chunk (injected: RELEARN_DQN_FAIL_CHUNK) must still leave network, Adam moments and step count as
    they were before the call — the update is all or nothing — so that the next update does exactly what
    it does for an agent that never saw the failure."""
    import os
    ref, _ = make(engine, n=256, capacity=64, minibatch=400, opt_steps=12)
    dqn, _ = make(engine, n=256, capacity=64, minibatch=400, opt_steps=12)
    ref.collect(40)
    dqn.collect(40)
    before = dqn.qnet.get_params()
    os.environ["RELEARN_DQN_FAIL_CHUNK"] = "2"  # chunks of 2, 4, 8 minibatches: six steps have run by then
    try:
        with pytest.raises(ra.RelearnError) as e:
            dqn.update()
        assert e.value.code == ra.ERR_INVALID_ARGUMENT
    finally:
        os.environ.pop("RELEARN_DQN_FAIL_CHUNK", None)
    assert np.array_equal(dqn.qnet.get_params(), before)
    # the failed call consumed its draws from the agent's Prng (as the reference's sampler would have before panicking);
    # give the reference agent the same position by letting it fail the same way, then both update identically
    os.environ["RELEARN_DQN_FAIL_CHUNK"] = "0"
    try:
        with pytest.raises(ra.RelearnError):
            ref.update()
    finally:
        os.environ.pop("RELEARN_DQN_FAIL_CHUNK", None)
    assert dqn.agent_rng_pos() == ref.agent_rng_pos()
    a, la = dqn.update(want_losses=True)
    b, lb = ref.update(want_losses=True)
    assert np.array_equal(la, lb) and np.array_equal(dqn.qnet.get_params(), ref.qnet.get_params())


def test_linear_schedule_and_collection_bound(engine):
    dqn, osim = make(engine, n=128, capacity=64, eps=("linear", 1.0, 0.1, 20000), minibatch=300, opt_steps=1)
    L = O.lib()
    cfgk = (1, 1.0, 0.1, 20000)
    assert dqn.exploration_rate(True) == L.oracle_exploration_rate(*cfgk, 0, 1) == 1.0
    assert dqn.exploration_rate(False) == 0.0
    b = L.oracle_collection_update_size(1, 1_000_000, 100_000, 0)
    assert dqn.min_update_size() == (b.min_steps, b.slack_steps)
    steps = 0
    for T in (40, 50):
        st = dqn.collect(T)
        eps_o = L.oracle_exploration_rate(*cfgk, steps, 1)
        assert st.exploration_rate == eps_o  # global_steps is refreshed by the update, as in dqn.rs:276
        flags_o, _ = osim.collect(T, eps_o)
        assert np.array_equal(dqn.replay_read(ra.REPLAY_LAST_FLAGS), flags_o)
        us = dqn.update()
        steps += T * dqn.n
        assert us.global_steps == steps
        osim.qparams[:] = dqn.qnet.get_params()  # keep the greedy branch of both sides on identical parameters
    assert dqn.exploration_rate(True) == L.oracle_exploration_rate(*cfgk, steps, 1) < 1.0


@pytest.mark.parametrize("td", [False, True], ids=["reward-to-go", "one-step-td"])
def test_sampling_and_targets_bit_exact(engine, td):
    dqn, osim = make(engine, n=300, capacity=64, minibatch=25000, td=td)
    for T in (40, 45):
        dqn.collect(T)
        osim.collect(T, 0.3)
    for k in range(4):
        sequential = k == 2  # the one-thread fallback path must give the same answer as the parallel one
        ne, ns = dqn.minibatch_sample(sequential=sequential)
        lanes, starts, lens, ns_o = osim.sample()
        assert (ne, ns) == (len(lanes), ns_o)
        assert ne > 1024  # more than one 1024-candidate chunk
        assert np.array_equal(dqn.minibatch_read(ra.MB_EP_LANE), lanes)
        assert np.array_equal(dqn.minibatch_read(ra.MB_EP_START), starts)
        assert np.array_equal(dqn.minibatch_read(ra.MB_EP_LEN), lens)
        assert np.array_equal(dqn.minibatch_read(ra.MB_EP_OFFSET), np.cumsum(lens) - lens)
        assert dqn.agent_rng_pos() == osim.agent_pos()
        obs_o, act_o, tgt_o = osim.minibatch(lanes, starts, lens)
        assert np.array_equal(dqn.minibatch_read(ra.MB_OBS).T, obs_o)
        assert np.array_equal(dqn.minibatch_read(ra.MB_ACTION), act_o.astype(np.uint8))
        assert np.array_equal(dqn.minibatch_read(ra.MB_TARGET), tgt_o)


@pytest.mark.parametrize("hidden,limit", [(128, ra.LIMIT_VISIBLE), (64, ra.LIMIT_VISIBLE), (128, ra.LIMIT_NONE)])
def test_minibatch_gradient(engine, variant, hidden, limit):
    dqn, osim = make(engine, n=256, hidden=hidden, capacity=96, minibatch=6000, limit=limit, td=True)
    dqn.collect(60)
    osim.collect(60, 0.3)
    dqn.minibatch_sample()
    lanes, starts, lens, _ = osim.sample()
    obs, act, tgt = osim.minibatch(lanes, starts, lens)
    g_d, loss_d = dqn.minibatch_gradient()
    g64, loss64 = osim.grad(obs, act, tgt, f64=True)
    g32, loss32 = osim.grad(obs, act, tgt)
    assert rel_err(g_d, g64) < GRAD_RTOL, (rel_err(g_d, g64), rel_err(g32, g64))
    assert abs(loss_d - loss64) <= 2e-6 * abs(loss64)
    assert np.isfinite(g_d).all() and np.abs(g_d).max() > 0


@pytest.mark.parametrize("td", [False, True], ids=["reward-to-go", "one-step-td"])
def test_update_against_oracle(engine, variant, td):
    dqn, osim = make(engine, n=256, capacity=96, minibatch=4000, td=td, opt_steps=6)
    dqn.collect(70)
    osim.collect(70, 0.3)
    st, losses_d = dqn.update(want_losses=True)
    losses_o = osim.update(6)
    assert st.opt_steps == 6 and st.global_steps == 70 * 256
    assert dqn.agent_rng_pos() == osim.agent_pos()  # the same episodes were drawn in every step
    assert np.max(np.abs(losses_d - losses_o) / np.abs(losses_o)) < 2e-5
    assert np.abs(dqn.qnet.get_params() - osim.qparams).max() < PARAM_ATOL


@pytest.mark.parametrize("limit", [ra.LIMIT_VISIBLE, ra.LIMIT_NONE], ids=["5-features", "4-features"])
def test_update_builds_all_minibatches_at_once(engine, limit):
    """Reward-to-go updates gather every minibatch of the update in one launch (and draw them on a second stream while
    the first ones train); the last one stays readable and is bit-identical to the oracle's (the same bars as the
    one-at-a-time builder), episodes longer than a wave's 64-step chunk included."""
    dqn, osim = make(engine, n=192, capacity=400, minibatch=9000, opt_steps=5, max_steps=150, eps=("const", 0.9),
                     limit=limit)
    dqn.collect(330)
    osim.collect(330, 0.9)
    st, losses_d = dqn.update(want_losses=True)
    for _ in range(4):
        osim.sample()
    lanes, starts, lens, ns = osim.sample()
    assert lens.max() > 64 and st.last_minibatch_steps == ns and st.last_minibatch_episodes == len(lanes)
    assert dqn.agent_rng_pos() == osim.agent_pos()
    obs_o, act_o, tgt_o = osim.minibatch(lanes, starts, lens)
    assert np.array_equal(dqn.minibatch_read(ra.MB_EP_LEN), lens)
    assert np.array_equal(dqn.minibatch_read(ra.MB_OBS).T, obs_o)
    assert np.array_equal(dqn.minibatch_read(ra.MB_ACTION), act_o.astype(np.uint8))
    assert np.array_equal(dqn.minibatch_read(ra.MB_TARGET), tgt_o)
    # and the one-at-a-time path takes the workspace back
    ne, ns2 = dqn.minibatch_sample()
    lanes, starts, lens, ns_o = osim.sample()
    assert (ne, ns2) == (len(lanes), ns_o)
    obs_o, act_o, tgt_o = osim.minibatch(lanes, starts, lens)
    assert np.array_equal(dqn.minibatch_read(ra.MB_OBS).T, obs_o)
    assert np.array_equal(dqn.minibatch_read(ra.MB_TARGET), tgt_o)


def test_cartpole_dqn_learns_something(engine):
    """A few collect/update rounds of the whole loop: the loss falls and nothing is NaN.  (The reference's own DQN
    test, learns_deterministic_bandit at dqn.rs:391-414, needs the bandit env; this is the CartPole analogue.)"""
    dqn, _ = make(engine, n=512, capacity=400, minibatch=20000, opt_steps=20, max_steps=200,
                  eps=("linear", 1.0, 0.1, 200000))
    first = last = None
    for it in range(4):
        dqn.collect(100)
        st = dqn.update()
        first = st.loss_first if first is None else first
        last = st.loss_last
        assert np.isfinite(st.loss_first) and np.isfinite(st.loss_last)
    assert last < first


# ------------------------------------------------------------------ action-value modules of any MlpConfig
# DqnConfig<MB> is generic over the module (src/torch/agents/dqn.rs:26-39): several hidden layers, other activations or
# a wider layer run the per-layer kernels (collection: one launch sequence per step).  Checked against DqnActor::act
# restated here on the raw actor stream (oracle Prng), the oracle's lanes for the env side, and the f64 NumPy network of
# tests/test_gpu_general_mlp.py.
def make_general(engine, hidden, act="Relu", n=64, capacity=64, minibatch=700, td=False, opt_steps=3, max_steps=23,
                 eps=0.3):
    env = ra.CartPoleEnv(engine, n, max_steps=max_steps, limit=ra.LIMIT_VISIBLE, seed_env=21, seed_actor=34)
    sim = O.LaneSim(n, max_steps=max_steps, limit=ra.LIMIT_VISIBLE, seed_env=21, seed_actor=34)
    q = ra.Mlp(engine, 5, hidden, 2, act, "Identity")
    q.init(77)
    cfg = ra.dqn_config_default()
    cfg.target = ra.DQN_TARGET_ONE_STEP_TD if td else ra.DQN_TARGET_REWARD_TO_GO
    cfg.exploration_kind, cfg.exploration_start = ra.SCHEDULE_CONSTANT, eps
    cfg.minibatch_steps, cfg.opt_steps_per_update, cfg.buffer_capacity, cfg.discount_factor = minibatch, opt_steps, capacity, 0.99
    for i, k in enumerate(KEY):
        cfg.agent_key[i] = k
    return ra.Dqn(env, q, ra.Adam(q), cfg), sim, q


@pytest.mark.parametrize("hidden,act", [([64, 64], "Relu"), ([200], "Relu"), ([32, 16, 8], "Tanh")])
def test_general_action_value_module_collection(engine, hidden, act):
    """every step of a collection: the env side replays bit for bit through the oracle's lanes, the explore / greedy
    decision and the random action are the lane's sequential actor-stream draws (Bernoulli, then gen_range(0..2)), the
    greedy action is the argmax of the module's outputs (compared where the two values are not within 1e-5), and the
    stream position the lane ends on is the oracle generator's"""
    import ctypes as C
    from test_gpu_general_mlp import forward64, unflatten
    n, T, eps = 64, 40, 0.3
    dqn, sim, q = make_general(engine, hidden, act, n=n, eps=eps)
    obs0 = sim.observe()
    dqn.collect(T)
    obs, act_rec, flag = dqn.replay_read(ra.REPLAY_OBS), dqn.replay_read(ra.REPLAY_ACTION), dqn.replay_read(ra.REPLAY_FLAG)
    nobs = dqn.replay_read(ra.REPLAY_NEXT_OBS)
    assert np.array_equal(dqn.replay_read(ra.REPLAY_TOTAL), np.full(n, T, dtype=np.uint32))
    net = unflatten(q.get_params(), 5, hidden, 2)
    L = O.lib()
    rngs = []
    for i in range(n):
        r = O.Prng()
        L.oracle_prng_seed_from_u64(C.byref(r), 34)
        L.oracle_prng_set_stream(C.byref(r), i)
        L.oracle_prng_set_word_pos(C.byref(r), 0)
        rngs.append(r)
    cur, greedy_checked, explored = obs0, 0, 0
    for t in range(T):
        assert np.array_equal(obs[:, t, :], cur), t  # slot t = step t (no eviction: T <= capacity)
        z, _ = forward64(net, cur.T, act, "Identity")
        for i in range(n):
            if L.oracle_prng_gen_bool(C.byref(rngs[i]), eps):
                assert act_rec[t, i] == L.oracle_prng_gen_range_u64(C.byref(rngs[i]), 0, 2), (t, i)
                explored += 1
            elif abs(z[i, 1] - z[i, 0]) > 1e-5:
                assert act_rec[t, i] == (1 if z[i, 1] > z[i, 0] else 0), (t, i)
                greedy_checked += 1
        reward, fl, nxt, term = sim.step(act_rec[t])
        want_flag = fl.copy()
        if t == T - 1:
            want_flag[fl == O.CONTINUE] = O.INTERRUPT  # the horizon rule: the open episode closes as Interrupt(successor)
        assert np.array_equal(flag[t], want_flag), t
        m = fl == O.INTERRUPT
        assert np.array_equal(nobs[:, t, m], term[:, m])
        cur = nxt
    assert explored > 0.2 * n * T and greedy_checked > 0.5 * n * T
    pos = np.array([L.oracle_prng_word_pos(C.byref(r)) for r in rngs], dtype=np.uint64)
    assert np.array_equal(dqn.replay_read(ra.REPLAY_ACTOR_POS), pos)


@pytest.mark.parametrize("td", [False, True], ids=["reward-to-go", "one-step-td"])
def test_general_action_value_module_update(engine, td):
    """a minibatch's targets and the MSE gradient on the taken action's value for a [64, 64] module against the f64
    NumPy network (one-step TD: r + gamma max_a Q(s'), 0 beyond a Terminate, successor = the next stored step or the
    stored Interrupt successor), then an update whose loss falls"""
    from test_gpu_general_mlp import backward64, forward64, unflatten
    hidden = [64, 64]
    dqn, sim, q = make_general(engine, hidden, td=td, n=128, capacity=96, minibatch=3000, opt_steps=8)
    dqn.collect(70)
    ne, ns = dqn.minibatch_sample()
    obs, a, tgt = dqn.minibatch_read(ra.MB_OBS), dqn.minibatch_read(ra.MB_ACTION), dqn.minibatch_read(ra.MB_TARGET)
    lanes, starts, lens = (dqn.minibatch_read(f) for f in (ra.MB_EP_LANE, ra.MB_EP_START, ra.MB_EP_LEN))
    net = unflatten(q.get_params(), 5, hidden, 2)
    robs, rflag, rnext = dqn.replay_read(ra.REPLAY_OBS), dqn.replay_read(ra.REPLAY_FLAG), dqn.replay_read(ra.REPLAY_NEXT_OBS)
    rrew = dqn.replay_read(ra.REPLAY_REWARD)
    want_t, k = np.zeros(ns), 0
    for ln, st, le in zip(lanes, starts, lens):
        slots = (int(st) + np.arange(int(le))) % dqn.C
        r = rrew[slots, ln].astype(np.float64)
        if td:
            succ = np.where((rflag[slots, ln] == O.INTERRUPT)[None] | (np.arange(le) == le - 1)[None],
                            rnext[:, slots, ln], robs[:, np.roll(slots, -1), ln])
            zn, _ = forward64(net, succ.T)
            vn = np.where(rflag[slots, ln] == O.TERMINATE, 0.0, zn.max(axis=1))
            want_t[k:k + le] = r + 0.99 * vn
        else:
            g = 0.0
            for j in range(int(le) - 1, -1, -1):
                g = r[j] + (0.99 * g if j < le - 1 else 0.0)
                want_t[k + j] = g
        k += int(le)
    assert k == ns and np.allclose(tgt, want_t, rtol=2e-5, atol=2e-5)
    z, acts = forward64(net, obs.T)
    qa = z[np.arange(ns), a]
    dz = np.zeros_like(z)
    dz[np.arange(ns), a] = 2.0 * (qa - tgt.astype(np.float64)) / ns
    want_g = backward64(net, obs.T, acts, dz)
    g_d, loss_d = dqn.minibatch_gradient()
    assert np.abs(g_d - want_g).max() <= 2e-5 * np.abs(want_g).max() + 1e-9
    assert abs(loss_d - ((qa - tgt) ** 2).mean()) <= 1e-5 * ((qa - tgt) ** 2).mean()
    p0 = q.get_params()
    st, losses = dqn.update(want_losses=True)
    assert st.opt_steps == 8 and np.all(np.isfinite(losses)) and not np.array_equal(q.get_params(), p0)
    if not td:  # (bootstrapped targets move with the network: eight steps on eight different minibatches need not lower it)
        assert losses[-1] < losses[0]
