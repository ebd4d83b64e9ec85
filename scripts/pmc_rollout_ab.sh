#!/bin/bash
# WRITE_SIZE of the fused rollout for two library builds (A/B of the lane -> XCD mapping): usage pmc_rollout_ab.sh <alt lib>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/ro_new -- python3 scripts/path_once.py 65536 3 4 > gpurun_out/pmc/ro_new.log 2>&1
RELEARN_LIB="$1" rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/ro_old -- python3 scripts/path_once.py 65536 3 4 > gpurun_out/pmc/ro_old.log 2>&1
python3 - <<PY
import csv, glob
for v in ("new", "old"):
    for f in glob.glob("gpurun_out/pmc/ro_%s/**/*counter_collection.csv" % v, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_rollout" in r["Kernel_Name"]]
        vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == "WRITE_SIZE"]
        print(v, len(vals), sum(vals) / max(len(vals), 1))
PY
