#!/bin/bash
# usage: scripts/prof_top.sh <python script> [args...]   (on the GPU box) — kernel-time summary of one run
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_top
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_top -o w -- python3 "$R/$1" "${@:2}" > $R/gpurun_out/prof_top.log 2>&1
python3 - <<PY
import csv,glob
fs=glob.glob("/tmp/prof_top/**/*kernel_stats.csv",recursive=True)
if not fs:
    print(open("$R/gpurun_out/prof_top.log").read()[-1500:])
else:
    for r in list(csv.DictReader(open(fs[0])))[:10]: print(r["Name"][:70],r["Calls"],r["AverageNs"],r["Percentage"])
PY
