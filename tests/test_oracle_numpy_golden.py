"""The oracle against independent double-entry goldens (tests/golden/make_numpy_golden.py: Python floats + platform
libm, written from the reference's source text, nothing of this repository included).  What these pin: the CartPole
single-step arithmetic the oracle shares a header with the kernels for (a defect in include/rl_detmath.h would be
invisible to every GPU-vs-oracle comparison), Chain transitions, the tabular-Q update rule.  CPU only."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O

L = O.lib()
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "numpy_golden.json")))


def _ulps(a, b):
    if a == b:
        return 0.0
    return abs(a - b) / np.spacing(max(abs(a), abs(b), 1e-300))


@pytest.mark.parametrize("use_libm", [0, 1], ids=["detmath", "libm"])
def test_cartpole_single_step_table(use_libm):
    env = O.CartPole()
    L.oracle_cartpole_default(C.byref(env))
    env.use_libm = use_libm
    worst = 0.0
    for case in G["cartpole"]:
        s = O.CartPoleState()
        s.x, s.xdot, s.th, s.thdot = case["state"]
        s.nv_pos = 1 if case["nv_pos"] else 0
        # next_state itself (the termination test of `step` discards the state)
        out = O.CartPoleState()
        force = -10.0 if case["action"] == 0 else 10.0
        L.oracle_cartpole_next_state(C.byref(env), C.byref(s), force, C.byref(out))
        want = [float.fromhex(h) for h in case["next"]]
        got = [out.x, out.xdot, out.th, out.thdot]
        for g, w in zip(got, want):
            # platform libm: bit for bit; the engine's own sincos is within 1 ulp of libm, which the physics after it
            # may stretch a little: two ulps of the result (absolute floor for results that cancel to ~0)
            if use_libm:
                assert g == w, (case["tag"], g, w)
            else:
                assert _ulps(g, w) <= 2 or abs(g - w) < 1e-17, (case["tag"], g, w)
                worst = max(worst, _ulps(g, w))
        assert bool(out.nv_pos) == case["next_nv_pos"], case["tag"]
        assert (bool(out.nv_pos) != case["nv_pos"]) == case["friction_flipped"], case["tag"]
        rew = C.c_double()
        succ = L.oracle_cartpole_step(C.byref(env), C.byref(s), case["action"], C.byref(rew))
        assert (succ == O.TERMINATE) == case["terminal"], case["tag"]
        assert rew.value == 1.0
    tags = {c["tag"] for c in G["cartpole"]}
    assert {"friction_flip", "pos_edge_out", "angle_edge_out", "negative_zero_velocity"} <= tags


def test_chain_transitions_on_recorded_draws():
    env = O.Chain()
    L.oracle_chain_default(C.byref(env))
    assert env.size == 5
    for row in G["chain"]:
        state = C.c_uint64(row["state"])
        rew = C.c_double()
        succ = L.oracle_chain_step_draw(C.byref(env), C.byref(state), row["action"], row["draw"], C.byref(rew))
        assert succ == O.CONTINUE and state.value == row["next"] and rew.value == row["reward"], row


def test_tabular_q_update_sequence():
    t = G["tabular_q"]
    q = L.oracle_tabular_q_new(t["n_obs"], t["n_act"], t["discount_factor"], 0.0)
    kinds = {"continue": O.CONTINUE, "terminate": O.TERMINATE, "interrupt": O.INTERRUPT}
    vals = np.zeros((t["n_obs"], t["n_act"]), np.float64)
    cnts = np.zeros((t["n_obs"], t["n_act"]), np.uint64)
    for st in t["steps"]:
        L.oracle_tabular_q_step_update(q, st["obs"], st["action"], st["reward"], kinds[st["next"]], st["next_obs"])
        L.oracle_tabular_q_read(q, O.f64p(vals), O.u64p(cnts))
        assert vals[st["obs"], st["action"]] == float.fromhex(st["q_after"]), st
        assert cnts[st["obs"], st["action"]] == st["count_after"]
    assert np.array_equal(vals, np.array([[float.fromhex(h) for h in row] for row in t["q_final"]]))
    assert np.array_equal(cnts, np.array(t["counts_final"], np.uint64))
    L.oracle_tabular_q_free(q)
