#!/bin/bash
# Copies what scripts/collect_profiles.sh left under gpurun_out/prof_<tag>/ into profiles/<tag>_* (the tracked evidence).
set -u
TAG="${1:-r05}"
SRC="gpurun_out/prof_$TAG"
DST="profiles"
for f in bench_65536 bench_32768 bench_16384 bench_8192 bench_4096 bench_under_rocprof_65536 bench_under_rocprof_8192 \
         bench_under_rocprof_4096 bench_under_rocprof_serial_65536 bench_under_rocprof_serial_8192 dqn_config3 gru_config5 \
         lstm_config5 general_mlp_period overlap_65536 overlap_8192 pmc_65536_summary pmc_gen_pair_summary pmc_dqn_summary \
         pmc_gru_config5_summary pmc_lstm_config5_summary stacked_gru_l2 stacked_lstm_l2 bench_65536_with_counters \
         rccl_one_rank_floor_8192; do
  [ -s "$SRC/$f.json" ] && cp "$SRC/$f.json" "$DST/${TAG}_$f.json"
done
cp "$SRC/general_mlp_passes.txt" "$DST/${TAG}_general_mlp_passes.txt" 2>/dev/null
for n in 65536 8192; do cp "$SRC/overlap_trace_$n.csv" "$DST/${TAG}_overlap_trace_$n.csv" 2>/dev/null; done
for n in 65536 8192 4096; do
  f=$(ls -t $(find "$SRC/stats_$n" -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp "$f" "$DST/${TAG}_bench_${n}_kernel_stats.csv"
done
for n in 65536 8192; do
  f=$(ls -t $(find "$SRC/stats_serial_$n" -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp "$f" "$DST/${TAG}_bench_${n}_serial_kernel_stats.csv"
done
f=$(ls -t $(find "$SRC/stats_dqn" -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp "$f" "$DST/${TAG}_dqn_config3_kernel_stats.csv"
ls "$DST" | grep "^${TAG}_" | wc -l
