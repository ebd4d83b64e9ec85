set -e
mkdir -p gpurun_out/lines
python3 bench.py > gpurun_out/lines/bench_65536.json
for n in 32768 16384 8192 4096; do python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/lines/bench_$n.json; done
python3 - <<PY
import json
for n in (65536,32768,16384,8192,4096):
    d=json.loads(open("gpurun_out/lines/bench_%d.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["value"]/1e6,1), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_us"],1), round(d["roofline"]["frac"],3), d["roofline"]["source"]["applies"], round(d["roofline_policy"]["avg_launch_us"],1), round(d["roofline_policy"]["frac"],3))
PY
