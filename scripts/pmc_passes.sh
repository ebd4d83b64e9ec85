#!/bin/bash
# PMC counter passes over one profiling target, each in its own rocprofv3 run (the counters do not fit one pass; the
# guide's HBM section wants FETCH_SIZE and WRITE_SIZE apart), with --kernel-trace only, the program directly after `--`.
#   usage: scripts/pmc_passes.sh <tag> <python script> [args...]      -> gpurun_out/pmc/<tag>/{sq_a,sq_b,fetch,write}/
# plus the same four passes over scripts/probe/fetch_calib (known byte counts) the first time, for the unit correction.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
TAG="$1"; shift
OUT="gpurun_out/pmc/$TAG"
rm -rf "$OUT" && mkdir -p "$OUT"
SQ_A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
SQ_B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
run_pass() {  # name, counters...
  local name="$1"; shift
  echo "[pmc_passes] $TAG/$name: $*"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 "${TARGET[@]}" > "$OUT/$name.log" 2>&1 \
    || { echo "[pmc_passes] pass $name FAILED, see $OUT/$name.log"; tail -5 "$OUT/$name.log"; return 1; }
}
TARGET=("$@")
run_pass sq_a $SQ_A && run_pass sq_b $SQ_B && run_pass fetch FETCH_SIZE && run_pass write WRITE_SIZE || exit 1
if [ ! -d gpurun_out/pmc/calib ] && [ -x scripts/probe/fetch_calib ]; then
  mkdir -p gpurun_out/pmc/calib
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/calib/fetch -- ./scripts/probe/fetch_calib > gpurun_out/pmc/calib/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc/calib/write -- ./scripts/probe/fetch_calib > gpurun_out/pmc/calib/write.log 2>&1
fi
python3 scripts/pmc_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
