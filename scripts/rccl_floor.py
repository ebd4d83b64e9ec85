#!/usr/bin/env python3
"""The measured floor under DESIGN 7a's multi-GPU projection: what the RCCL transport costs a rank BEFORE any link is
involved.  One rank with a real RCCL communicator (RELEARN_FORCE_RCCL=1: dlopen, ncclCommInitRank, both communicators,
ncclAllReduce on the engine's streams) against the same rank without a collective, at one rank's share of config 4
(8,192 lanes, T = 128, 80 critic steps):
  - per call: the HIP-event span of an ncclAllReduce on a <= 4 KiB vector (class `allreduce`),
  - per critic step: reduce -> all-reduce -> Adam (three launches) against the fused reduce + Adam launch,
  - per period: the difference of the two period times, chains in turn and side by side.
A one-rank all-reduce moves nothing over xGMI: a real N-rank call adds the link round trips on top of this.
usage: rccl_floor.py [lanes] [periods] [out.json]   (RCCL prints a banner on stdout: give a file for the JSON object)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RELEARN_FORCE_RCCL"] = "1"
import relearn_amd as ra  # noqa: E402

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
periods = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, H = 128, 128


def run(with_comm, serial, profile):
    eng = ra.Engine(0)
    if with_comm:
        eng.comm_init(0, 1, ra.comm_unique_id())
    env = ra.CartPoleEnv(eng, lanes, max_steps=500, seed_env=0, seed_actor=1)
    pol, cri = ra.Mlp(eng, 5, H, 2), ra.Mlp(eng, 5, H, 1)
    pol.init(2)
    cri.init(3)
    opt = ra.Adam(cri)
    traj = ra.Trajectory(eng, lanes, T, 5)
    eng.set_serial_update(serial)

    def period():
        ra.rollout(env, pol, traj)
        ra.gae(traj, cri, 0.99, 0.95)
        ra.actor_critic_update(pol, cri, opt, traj)

    for _ in range(3):
        period()
    eng.sync()
    eng.profile_read(reset=True)
    eng.profile_enable(profile)
    t0 = time.perf_counter()
    for _ in range(periods):
        period()
    eng.sync()
    ms = 1e3 * (time.perf_counter() - t0) / periods
    prof = eng.profile_read(reset=True) if profile else None
    eng.profile_enable(False)
    if with_comm:
        eng.comm_destroy()
    out = {"ms_per_period": ms}
    if prof:
        out["classes"] = {k: {"launches_per_period": v[1] / periods, "us_per_launch": 1e3 * v[0] / max(v[1], 1)}
                          for k, v in prof.items() if v[1]}
    return out


res = {"lanes": lanes, "horizon": T, "critic_steps": 80, "periods": periods,
       "rccl": dict(zip(("librccl", "libamdhip64"), ra.comm_library_paths())) if ra.comm_available() else None}
for name, with_comm in (("no_collective", False), ("rccl_one_rank", True)):
    res[name] = {"side_by_side": run(with_comm, False, False), "in_turn": run(with_comm, True, False),
                 "in_turn_profiled": run(with_comm, True, True)}
a, b = res["no_collective"], res["rccl_one_rank"]
cls = b["in_turn_profiled"]["classes"]
n_ar = cls.get("allreduce", {}).get("launches_per_period", 0.0)
res["floor"] = {
    "allreduce_calls_per_period": n_ar,
    "allreduce_event_span_us": cls.get("allreduce", {}).get("us_per_launch"),
    "added_ms_per_period_in_turn": b["in_turn"]["ms_per_period"] - a["in_turn"]["ms_per_period"],
    "added_ms_per_period_side_by_side": b["side_by_side"]["ms_per_period"] - a["side_by_side"]["ms_per_period"],
    "added_us_per_allreduce_in_turn": 1e3 * (b["in_turn"]["ms_per_period"] - a["in_turn"]["ms_per_period"]) / max(n_ar, 1),
    "note": "one rank: ncclAllReduce has no peer to wait for, so this is the cost of the call path and of the unfused "
            "reduce -> all-reduce -> Adam sequence only; link latency at 2-8 ranks comes on top",
}
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
else:
    print(json.dumps(res, indent=1))
