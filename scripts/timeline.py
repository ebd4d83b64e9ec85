#!/usr/bin/env python3
"""Per-queue timeline of one steady-state period of bench.py from a rocprofv3 --kernel-trace: every launch between two
consecutive k_rollout_cartpole starts (the second-to-last pair of the trace), with its hardware queue, start and duration,
plus the idle gaps of each queue.
usage: timeline.py <dir with *_kernel_trace.csv> <out.csv> [period index from the end, default 2]"""
import csv
import glob
import json
import os
import sys

src, out_csv = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = []
for f in glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"].split("(")[0][:48]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_rollout_cartpole" in r[3]]
# a period in the accounting sense: from the first launch after rollout k's GAE ... simpler: rollout k start -> rollout k+1 start
lo, hi = starts[-back - 1], starts[-back]
t0 = rows[lo][0]
seg = [r for r in rows if r[0] >= t0 and r[0] < rows[hi][0] + 1]
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "queue", "start_us", "dur_us"])
    for s, e, q, k in seg:
        w.writerow([k, q, "%.2f" % ((s - t0) / 1e3), "%.2f" % ((e - s) / 1e3)])
per_q = {}
for s, e, q, k in seg:
    d = per_q.setdefault(q, {"launches": 0, "busy_us": 0.0, "first_us": (s - t0) / 1e3, "last_end_us": 0.0})
    d["launches"] += 1
    d["busy_us"] += (e - s) / 1e3
    d["last_end_us"] = max(d["last_end_us"], (e - t0) / 1e3)
print(json.dumps({"rollout_to_rollout_us": (rows[hi][0] - t0) / 1e3, "queues": per_q}, indent=1))
