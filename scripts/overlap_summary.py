#!/usr/bin/env python3
"""What runs beside what in a rocprofv3 --kernel-trace of bench.py: the launches of the two update chains of
rl_actor_critic_update (two HIP streams = two hardware queues) with their start / end stamps, and how much of each
chain's time another queue's kernel was running.
usage: overlap_summary.py <dir with *_kernel_trace.csv> <out.csv> [<out.json>]"""
import csv
import glob
import json
import os
import sys

src, out_csv = sys.argv[1], sys.argv[2]
out_json = sys.argv[3] if len(sys.argv) > 3 else None
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
# the last period with critic steps on a queue of their own: from the last k_rollout launch on
starts = [i for i, r in enumerate(rows) if "k_rollout_cartpole" in r[3]]
crit_q = {}
for s, e, q, k in rows:
    if "k_critic_step" in k:
        crit_q[q] = crit_q.get(q, 0) + 1
# a period run side by side: the critic steps' queue differs from the rollout's
pick = None
for i in reversed(range(len(starts))):
    lo = starts[i]
    hi = starts[i + 1] if i + 1 < len(starts) else len(rows)
    seg = rows[lo:hi]
    qs = {q for _, _, q, k in seg if "k_critic_step" in k} | {q for _, _, q, k in seg if "k_policy_bf16" in k}
    if len(qs) >= 2:
        pick = seg
        break
if pick is None:
    pick = rows[starts[-1]:] if starts else rows
t0 = pick[0][0]
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "queue", "start_us", "end_us", "dur_us", "other_queue_busy_us"])
    tot = {}
    for s, e, q, k in pick:
        other = 0
        for s2, e2, q2, _ in pick:
            if q2 != q and s2 < e and e2 > s:
                other += min(e, e2) - max(s, s2)
        w.writerow([k, q, "%.2f" % ((s - t0) / 1e3), "%.2f" % ((e - t0) / 1e3), "%.2f" % ((e - s) / 1e3), "%.2f" % (other / 1e3)])
        d = tot.setdefault(q, {"launches": 0, "busy_us": 0.0, "beside_other_queue_us": 0.0})
        d["launches"] += 1
        d["busy_us"] += (e - s) / 1e3
        d["beside_other_queue_us"] += min(other, e - s) / 1e3
span = (max(r[1] for r in pick) - t0) / 1e3
summary = {"period_span_us": span, "queues": tot, "launches": len(pick),
           "note": "one period of bench.py under rocprofv3 --kernel-trace; queue = hardware queue of the HIP stream "
                   "(main stream: rollout, values, GAE, TRPO chain; auxiliary stream: critic chain)"}
print(json.dumps(summary, indent=1))
if out_json:
    json.dump(summary, open(out_json, "w"), indent=1)
