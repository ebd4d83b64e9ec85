#!/bin/bash
# Collects the round's evidence on a GPU box into gpurun_out/prof_<tag>/ (copy what is to be judged into profiles/):
#   bench lines at 65,536 / 8,192 / 4,096 lanes, the rocprofv3 --kernel-trace --stats summary of the bench command,
#   and the PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE) over one period of the hot path.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:-r05}"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT="gpurun_out/prof_$TAG"
PART="${2:-all}"   # part1: the headline configuration (bench lines, kernel traces, counters); part2: the other configurations
[ "$PART" = "part2" ] || rm -rf "$OUT"
mkdir -p "$OUT"
if [ "$PART" != "part2" ]; then
python3 bench.py > "$OUT/bench_65536.json" 2> "$OUT/bench_65536.err" || echo "bench 65536 failed"
# one rank's share of the headline at 2 / 4 / 8 / 16 GPUs (the projection is built from these)
for n in 32768 16384 8192 4096; do
  python3 bench.py --envs $n --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err" || echo "bench $n failed"
done
for n in 65536 8192 4096; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$n" -- python3 bench.py --envs $n --steps 3 --warmup 1 --no-cpu-baseline \
    > "$OUT/bench_under_rocprof_$n.json" 2> "$OUT/stats_$n.log" || echo "rocprof stats $n failed"
done
# the same command with the two update chains in turn: per-kernel averages that agree with the HIP-event figures of the
# bench line (its profiled period runs the chains in turn; side by side, a launch's span includes the other chain's work)
for n in 65536 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_serial_$n" -- python3 bench.py --envs $n --steps 3 --warmup 1 --no-cpu-baseline --serial-update \
    > "$OUT/bench_under_rocprof_serial_$n.json" 2> "$OUT/stats_serial_$n.log" || echo "rocprof serial stats $n failed"
done
bash scripts/pmc_passes.sh "${TAG}_65536" scripts/path_once.py 65536 2 80 > "$OUT/pmc_65536.log" 2>&1 || echo "pmc failed"
cp "gpurun_out/pmc/${TAG}_65536/summary.json" "$OUT/pmc_65536_summary.json" 2>/dev/null
# (the measurement scripts below read the counter summaries from profiles/ and use them when they come from this library)
cp "$OUT/pmc_65536_summary.json" "profiles/${TAG}_pmc_65536_summary.json" 2>/dev/null
# which launches of the two update chains ran side by side (the non-profiled periods of the traced bench run)
python3 scripts/overlap_summary.py "$OUT/stats_65536" "$OUT/overlap_trace_65536.csv" "$OUT/overlap_65536.json" > /dev/null 2>&1 || echo "overlap 65536 failed"
python3 scripts/overlap_summary.py "$OUT/stats_8192" "$OUT/overlap_trace_8192.csv" "$OUT/overlap_8192.json" > /dev/null 2>&1 || echo "overlap 8192 failed"
python3 bench.py > "$OUT/bench_65536_with_counters.json" 2> "$OUT/bench_65536_with_counters.err" || echo "bench (with counters) failed"
python3 scripts/rccl_floor.py 8192 20 "$OUT/rccl_one_rank_floor_8192.json" > "$OUT/rccl_floor.log" 2>&1 || echo "rccl floor failed"
fi
[ "$PART" = "part1" ] && { echo "collect_profiles part1 done"; exit 0; }
# counters of the other configurations' dominant kernels: the pair kernel of a [64, 64] critic, the DQN update, the GRU passes
bash scripts/pmc_passes.sh "${TAG}_gen_pair" scripts/gen_critic_only.py 16384 10 64 64 > "$OUT/pmc_gen_pair.log" 2>&1 || echo "pmc gen failed"
cp "gpurun_out/pmc/${TAG}_gen_pair/summary.json" "$OUT/pmc_gen_pair_summary.json" 2>/dev/null
bash scripts/pmc_passes.sh "${TAG}_dqn" scripts/dqn_config3.py 4096 1221 3 > "$OUT/pmc_dqn.log" 2>&1 || echo "pmc dqn failed"
cp "gpurun_out/pmc/${TAG}_dqn/summary.json" "$OUT/pmc_dqn_summary.json" 2>/dev/null
cp "$OUT/pmc_dqn_summary.json" "profiles/${TAG}_pmc_dqn_summary.json" 2>/dev/null
bash scripts/pmc_passes.sh "${TAG}_gru" scripts/gru_config5.py 16384 100 8 1 > "$OUT/pmc_gru.log" 2>&1 || echo "pmc gru failed"
cp "gpurun_out/pmc/${TAG}_gru/summary.json" "$OUT/pmc_gru_config5_summary.json" 2>/dev/null
cp "$OUT/pmc_gru_config5_summary.json" "profiles/${TAG}_pmc_gru_config5_summary.json" 2>/dev/null
bash scripts/pmc_passes.sh "${TAG}_lstm" scripts/lstm_config5.py 16384 100 8 1 > "$OUT/pmc_lstm.log" 2>&1 || echo "pmc lstm failed"
cp "gpurun_out/pmc/${TAG}_lstm/summary.json" "$OUT/pmc_lstm_config5_summary.json" 2>/dev/null
# the other configurations: DQN (config 3) with its kernel trace, GRU (config 5), the general MLP period and its passes
python3 scripts/dqn_config3.py > "$OUT/dqn_config3.json" 2> "$OUT/dqn_config3.err" || echo "dqn failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_dqn" -- python3 scripts/dqn_config3.py 4096 1221 3 \
  > "$OUT/dqn_under_rocprof.json" 2> "$OUT/stats_dqn.log" || echo "rocprof dqn failed"
python3 scripts/lstm_config5.py > "$OUT/lstm_config5.json" 2> "$OUT/lstm_config5.err" || echo "lstm failed"
python3 scripts/gru_config5.py > "$OUT/gru_config5.json" 2> "$OUT/gru_config5.err" || echo "gru failed"
python3 scripts/general_mlp_period.py > "$OUT/general_mlp_period.json" 2> "$OUT/general_mlp_period.err" || echo "general failed"
python3 scripts/gen_passes.py > "$OUT/general_mlp_passes.txt" 2>&1 || echo "gen passes failed"
# stacked recurrent layers on the lane-per-thread kernels (2-layer GRU / LSTM of width 128, 4,096 lanes x 100 steps)
python3 scripts/stacked_period.py 4096 100 128 2 gru > "$OUT/stacked_gru_l2.json" 2> "$OUT/stacked_gru_l2.err" || echo "stacked gru failed"
python3 scripts/stacked_period.py 4096 100 128 2 lstm > "$OUT/stacked_lstm_l2.json" 2> "$OUT/stacked_lstm_l2.err" || echo "stacked lstm failed"
find "$OUT" -name "*kernel_stats.csv" | head
echo "collect_profiles done"
