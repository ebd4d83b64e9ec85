#!/usr/bin/env python3
"""Independent double-entry goldens for the pieces the oracle and the kernels SHARE a header for (include/rl_detmath.h,
include/rl_chacha.h): written from the reference's source text with Python floats (IEEE f64) and the platform libm
(math.sin / math.cos), without including or calling anything of this repository.

  cartpole   one step of Wrapped<CartPole, ...>'s inner env: InternalPhysicalConstants::{next_state,
             angular_acceleration, normal_force} (src/envs/cartpole.rs:306-446), constants (:178-216, :238-251), the
             termination rule of CartPole::step (:128-154) — 64 random states x 2 actions, a friction-sign flip, both
             termination edges, a -0.0 cart velocity
  chain      Chain::step (src/envs/chain.rs:83-105) on a recorded list of slip draws
  tabular_q  BaseTabularQLearningAgent::step_update (src/agents/tabular.rs:159-180): a 20-step sequence

usage: python tests/golden/make_numpy_golden.py   -> tests/golden/numpy_golden.json
"""
import json
import math
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))

# PhysicalConstants::default / EnvironmentParams::default (cartpole.rs:178-216)
GRAVITY, MASS_CART, MASS_POLE, HALF_LEN = 9.8, 1.0, 0.1, 0.5
FRICTION_CART, FRICTION_POLE, TIME_STEP = 0.01, 0.01, 0.02
ACTION_FORCE, MAX_POS, MAX_ANGLE = 10.0, 2.4, math.radians(12.0)
# From<PhysicalConstants> for InternalPhysicalConstants (cartpole.rs:238-251)
TOTAL_MASS = MASS_CART + MASS_POLE
TOTAL_WEIGHT = GRAVITY * TOTAL_MASS
INV_TOTAL_MASS = 1.0 / TOTAL_MASS
MASS_LENGTH_POLE = MASS_POLE * HALF_LEN


def angular_acceleration(thdot, applied_force, signed_cart_friction, w2, sin_a, cos_a):
    alpha = (-applied_force - MASS_LENGTH_POLE * w2 * (sin_a + signed_cart_friction * cos_a)) * INV_TOTAL_MASS
    beta = FRICTION_POLE * thdot / MASS_LENGTH_POLE
    numerator = GRAVITY * sin_a + cos_a * (alpha + GRAVITY * signed_cart_friction) - beta
    denominator = HALF_LEN * (4.0 / 3.0 - MASS_POLE * cos_a * INV_TOTAL_MASS * (cos_a - signed_cart_friction))
    return numerator / denominator


def normal_force(acc, w2, sin_a, cos_a):
    return TOTAL_WEIGHT - MASS_LENGTH_POLE * (acc * sin_a + w2 * cos_a)


def is_sign_positive(v):
    return math.copysign(1.0, v) > 0.0  # +0.0 counts as positive, -0.0 as negative


def next_state(x, xdot, th, thdot, nv_pos, applied_force):
    signed = FRICTION_CART if nv_pos else -FRICTION_CART
    sin_a, cos_a = math.sin(th), math.cos(th)
    w2 = thdot * thdot
    acc = angular_acceleration(thdot, applied_force, signed, w2, sin_a, cos_a)
    nf = normal_force(acc, w2, sin_a, cos_a)
    new_pos = is_sign_positive(nf * xdot)
    flipped = new_pos != nv_pos
    if flipped:
        signed = -signed
        acc = angular_acceleration(thdot, applied_force, signed, w2, sin_a, cos_a)
        nf = normal_force(acc, w2, sin_a, cos_a)
    force_pole = MASS_LENGTH_POLE * (w2 * sin_a + acc * cos_a)
    force_friction = -signed * nf
    net = applied_force + force_pole + force_friction
    cart_acc = net * INV_TOTAL_MASS
    xdot2 = xdot + TIME_STEP * cart_acc
    x2 = x + TIME_STEP * xdot2          # the NEW velocity (semi-implicit Euler)
    thdot2 = thdot + TIME_STEP * acc
    th2 = th + TIME_STEP * thdot        # the OLD angular velocity
    return (x2, xdot2, th2, thdot2, new_pos), flipped


def cartpole_step(state, action):
    """CartPole::step: Push::Left = 0 -> -force; terminal iff |x'| > max_pos or |theta'| > max_angle (strict)"""
    force = -ACTION_FORCE if action == 0 else ACTION_FORCE
    nxt, flipped = next_state(*state, force)
    terminal = abs(nxt[0]) > MAX_POS or abs(nxt[2]) > MAX_ANGLE
    return nxt, terminal, flipped


def cartpole_table():
    rnd = random.Random(20261004)
    cases = []

    def add(tag, state, action):
        nxt, term, flipped = cartpole_step(state, action)
        cases.append({"tag": tag, "state": [float(v) for v in state[:4]], "nv_pos": bool(state[4]), "action": action,
                      "next": [nxt[0].hex(), nxt[1].hex(), nxt[2].hex(), nxt[3].hex()], "next_nv_pos": bool(nxt[4]),
                      "terminal": bool(term), "friction_flipped": bool(flipped)})

    for i in range(64):
        st = (rnd.uniform(-2.3, 2.3), rnd.uniform(-2.0, 2.0), rnd.uniform(-0.2, 0.2), rnd.uniform(-2.5, 2.5),
              rnd.random() < 0.8)
        for a in (0, 1):
            add("random%02d" % i, st, a)
    # the cached sign disagrees with the cart's direction: the friction sign flips and the step is recomputed once
    add("friction_flip", (0.1, -0.7, 0.03, 0.4, True), 1)
    add("friction_flip_back", (0.1, 0.7, 0.03, 0.4, False), 0)
    # termination edges: just inside / just outside the position and the angle bound (strict >)
    add("pos_edge_in", (2.38, 0.2, 0.0, 0.0, True), 0)
    add("pos_edge_out", (2.395, 0.5, 0.0, 0.0, True), 1)
    add("pos_edge_neg_out", (-2.395, -0.5, 0.0, 0.0, False), 0)
    add("angle_edge_in", (0.0, 0.0, MAX_ANGLE - 0.02, 0.1, True), 0)
    add("angle_edge_out", (0.0, 0.0, MAX_ANGLE - 0.001, 0.5, True), 1)
    # -0.0 velocity: normal_force * -0.0 = -0.0 -> is_sign_positive false
    add("negative_zero_velocity", (0.0, -0.0, 0.01, 0.0, True), 1)
    add("positive_zero_velocity", (0.0, 0.0, 0.01, 0.0, True), 1)
    assert any(c["friction_flipped"] for c in cases) and any(c["terminal"] for c in cases)
    return cases


def chain_table():
    """Chain::default(): 5 states; `if rng.gen::<f32>() < 0.2 { action = action.invert() }`; Left -> (0, 2.0);
    Right -> (s + 1, 0.0) or, at the last state, (s, 10.0)"""
    rnd = random.Random(7)
    size = 5
    rows, state = [], 0
    for i in range(60):
        draw = [0.0, 0.19999999, 0.2, 0.20000002, 0.5, 0.999][i % 6] if i < 12 else rnd.random()
        import struct
        draw = struct.unpack("f", struct.pack("f", draw))[0]  # an f32 value
        action = rnd.randrange(2) if i % 5 else 1
        a = action
        if draw < struct.unpack("f", struct.pack("f", 0.2))[0]:
            a = 1 - a
        if a == 0:
            nxt, rew = 0, 2.0
        elif state == size - 1:
            nxt, rew = state, 10.0
        else:
            nxt, rew = state + 1, 0.0
        rows.append({"state": state, "action": action, "draw": draw, "next": nxt, "reward": rew})
        state = nxt
    assert any(r["reward"] == 10.0 for r in rows)
    return rows


def tabular_q_sequence():
    """step_update: next value = max_a Q[next_obs, a] * discount (0 when the episode terminated); counts += 1;
    weight = 1 / count; Q *= 1 - weight; Q += weight * (reward + next value)"""
    n_obs, n_act, gamma = 3, 2, 0.9
    q = [[0.0] * n_act for _ in range(n_obs)]
    cnt = [[0] * n_act for _ in range(n_obs)]
    rnd = random.Random(11)
    steps = []
    for i in range(20):
        obs, act = rnd.randrange(n_obs), rnd.randrange(n_act)
        reward = float(rnd.randrange(-3, 8))
        kind = ["continue", "terminate", "interrupt"][rnd.randrange(3)] if i >= 2 else "terminate"
        nobs = rnd.randrange(n_obs)
        nv = 0.0 if kind == "terminate" else max(q[nobs]) * gamma
        cnt[obs][act] += 1
        w = 1.0 / float(cnt[obs][act])
        q[obs][act] *= 1.0 - w
        q[obs][act] += w * (reward + nv)
        steps.append({"obs": obs, "action": act, "reward": reward, "next": kind, "next_obs": nobs,
                      "q_after": q[obs][act].hex(), "count_after": cnt[obs][act]})
    return {"n_obs": n_obs, "n_act": n_act, "discount_factor": gamma, "steps": steps,
            "q_final": [[v.hex() for v in row] for row in q], "counts_final": cnt}


def main():
    data = {"generator": "tests/golden/make_numpy_golden.py (Python floats + platform libm; no code of this repository)",
            "cartpole": cartpole_table(), "chain": chain_table(), "tabular_q": tabular_q_sequence(),
            # the two values tests/test_oracle_lanes.py used to compute without asserting them: Q(0,1) after
            # (reward 4, terminate) and then (reward 2, terminate) with the 1/n step size
            "tabular_q_two_steps": [4.0, 3.0]}
    with open(os.path.join(HERE, "numpy_golden.json"), "w") as f:
        json.dump(data, f, indent=1)
    print("wrote numpy_golden.json: %d cartpole cases, %d chain rows, %d tabular steps" % (
        len(data["cartpole"]), len(data["chain"]), len(data["tabular_q"]["steps"])))


if __name__ == "__main__":
    main()
