#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (no trace domains mixed in), calibrated on a known byte count
# with the same access shape (4 B per lane, coalesced).  Writes gpurun_out/traffic/r01_traffic.json.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/traffic
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cal_f -- ./scripts/probe/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cal_w -- ./scripts/probe/fetch_calib > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/k_f -- python3 scripts/critic_only.py 65536 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/k_w -- python3 scripts/critic_only.py 65536 4 > /dev/null 2>&1
python3 scripts/traffic_report.py
