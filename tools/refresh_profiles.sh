#!/bin/bash
# Re-collects everything under profiles/ that DESIGN.md and bench.py quote, for the current build (round tag below).
# Run on the GPU box from the repo root (about 6 minutes):  bash tools/refresh_profiles.sh gpurun_out/prof
# Afterwards copy the r03_* files it lists from gpurun_out/prof/export/ into profiles/ and commit them.
set -e
R=r03
OUT=${1:-gpurun_out/prof}; mkdir -p "$OUT" "$OUT/export"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
# 1. HBM traffic (PMC) first: bench.py reads the bench-batch figure from profiles/
bash tools/collect_traffic.sh "$OUT/traffic_bench" > "$OUT/traffic_bench.log" 2>&1
cp "$OUT/traffic_bench/traffic.json" "$OUT/export/${R}_pmc_traffic_bench.json"
cp "$OUT/traffic_bench/traffic.json" "profiles/${R}_pmc_traffic_bench.json"
PROBE=1 bash tools/collect_traffic.sh "$OUT/traffic_c5" > "$OUT/traffic_c5.log" 2>&1
cp "$OUT/traffic_c5/traffic.json" "$OUT/export/${R}_pmc_traffic_c5x32_binning.json"
# 2. instruction / wait counters of every kernel
bash tools/collect_sq.sh "$OUT/sq" > "$OUT/sq.log" 2>&1
( echo "# rocprofv3 --pmc instruction counters, mean per dispatch (tools/collect_sq.sh: bench.py --streams 1 --steps 3, 2048 frames = 258.8 M points per dispatch)"; grep -v debug "$OUT/sq/summary.txt" ) > "$OUT/export/${R}_sq_counters.txt"
# 3. the default bench line (with host path, latency and CPU baseline), the other workloads, variant B, end to end
python3 bench.py > "$OUT/export/${R}_bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --no-cpu-baseline --no-host-path --end-to-end > "$OUT/export/${R}_bench_end_to_end.json" 2> "$OUT/bench_e2e.err"
for w in c1 rough3 rough6; do
  python3 bench.py --workload $w --steps 12 --warmup 3 --no-host-path > "$OUT/export/${R}_bench_workload_$w.json" 2> "$OUT/bench_$w.err"
done
for w in c4 c5; do      # with the host-path legs: these lines also carry `four_batches_in_flight` (few, long patches per wave: it matters here)
  python3 bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline > "$OUT/export/${R}_bench_workload_$w.json" 2> "$OUT/bench_$w.err"
done
for w in c2 c4 c5; do
  python3 bench.py --workload $w --max-iter 0 --steps 12 --warmup 3 --no-host-path --no-cpu-baseline > "$OUT/export/${R}_bench_variantB_$w.json" 2> "$OUT/bench_vb_$w.err"
done
# 4. rocprofv3 kernel statistics of the same default command (without the extra single-frame / host-path launches, so
#    that the averages describe the bench batch), and of the binning probe
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_bench" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-host-path > "$ROOT/$OUT/export/${R}_bench_under_rocprof.json" 2> "$ROOT/$OUT/stats_bench.err" )
( cd /tmp && export TMPDIR=/tmp MAX_ITER=1 && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_c5" -o t -- python3 "$ROOT/tools/binning_probe.py" > "$ROOT/$OUT/export/${R}_binning_probe_c5x32.json" 2> "$ROOT/$OUT/stats_c5.err" )
cp "$(find "$OUT/stats_bench" -name '*kernel_stats.csv' | head -1)" "$OUT/export/${R}_bench_kernel_stats.csv"
cp "$(find "$OUT/stats_c5" -name '*kernel_stats.csv' | head -1)" "$OUT/export/${R}_binning_probe_c5x32_kernel_stats.csv"
# launches of a kernel before the timed region: 1 set-up + warm-up (6); timed: 56
python3 tools/trace_phase_avg.py "$(find "$OUT/stats_bench" -name '*kernel_trace.csv' | head -1)" 7 56 > "$OUT/export/${R}_bench_kernel_trace_summary.json"
# 5. the loop micro-benchmark and the host path probe
python3 tools/loop_bench.py 8192 > "$OUT/export/${R}_loop_bench.txt" 2>/dev/null
python3 tools/loop_form_probe.py 8192 > "$OUT/export/${R}_loop_form_probe.txt" 2>/dev/null
python3 tools/loop_pct_probe.py 8192 > "$OUT/export/${R}_loop_pct_probe.txt" 2>/dev/null
python3 tools/host_path_probe.py --slots 1,2,4,8,16 > "$OUT/export/${R}_host_path_probe.json" 2>/dev/null
ls "$OUT/export"
tail -c 400 "$OUT/export/${R}_bench_default.json"
