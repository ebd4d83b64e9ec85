#!/bin/bash
# A/B of rl_actor_critic_update: the two update chains side by side on two streams against one after the other, at the
# headline lane count and at one rank's share of it.  Output: gpurun_out/ab/*.json + a table on stdout.
set -e
mkdir -p gpurun_out/ab
for n in 65536 8192 4096; do
  python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/ab/overlap_$n.json
  python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline --serial-update > gpurun_out/ab/serial_$n.json
  python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-profile > gpurun_out/ab/overlap_noprof_$n.json
done
python3 - <<PY
import json
for n in (65536, 8192, 4096):
    row = [n]
    for k in ("overlap", "serial", "overlap_noprof"):
        d = json.loads(open("gpurun_out/ab/%s_%d.json" % (k, n)).read().strip().splitlines()[-1])
        row += [k, round(d["ms_per_step"], 3), round(d["value"] / 1e6, 1)]
    print(*row)
PY
