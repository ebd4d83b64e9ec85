#!/bin/bash
# A/B on one box: this round's library against round 5's (scripts/probe/abl/librelearn_r05.so, built from the round-5
# sources: same sha as profiles/r05_*), the period at the headline lane count and at one rank's share of it, in turn and
# twice each.  Output: gpurun_out/ab6/*.json + a table on stdout.
set -e
mkdir -p gpurun_out/ab6
for rep in 1 2; do
for n in 8192 4096 65536; do
  RELEARN_LIB=scripts/probe/abl/librelearn_r05.so python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab6/r05_${n}_$rep.json
  python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab6/r06_${n}_$rep.json
done
done
python3 - <<PY
import json
for n in (65536, 8192, 4096):
    row = [n]
    for k in ("r05", "r06"):
        for rep in (1, 2):
            d = json.loads(open("gpurun_out/ab6/%s_%d_%d.json" % (k, n, rep)).read().strip().splitlines()[-1])
            row += ["%s.%d" % (k, rep), round(d["ms_per_step"], 3), round(d["roofline"]["avg_launch_us"], 2)]
    print(*row)
PY
