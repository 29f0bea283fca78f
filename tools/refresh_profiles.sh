#!/bin/bash
# Re-collects everything under profiles/ that DESIGN.md and bench.py quote, for the current build.
# Run on the GPU box from the repo root (about 4 minutes):  bash tools/refresh_profiles.sh gpurun_out/prof
set -e
OUT=${1:-gpurun_out/prof}; mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
# 1. HBM traffic (PMC) first: bench.py reads the bench-batch figure from profiles/
bash tools/collect_traffic.sh "$OUT/traffic_bench" > "$OUT/traffic_bench.log" 2>&1
PROBE=1 bash tools/collect_traffic.sh "$OUT/traffic_c5" > "$OUT/traffic_c5.log" 2>&1
cp "$OUT/traffic_bench/traffic.json" profiles/r01_pmc_traffic_bench.json
# 2. the default bench line (with the CPU baseline) and the end-to-end / single-frame variants
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --no-cpu-baseline --end-to-end --single-frame-latency > "$OUT/bench_e2e.json" 2> "$OUT/bench_e2e.err"
python3 bench.py --no-cpu-baseline --frames 1 --streams 1 --steps 200 --warmup 20 > "$OUT/bench_single_frame.json" 2> "$OUT/bench_single.err"
# 3. rocprofv3 kernel statistics of the same default command, and of the binning probe
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_bench" -o t -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/stats_bench.err" )
( cd /tmp && export TMPDIR=/tmp MAX_ITER=1 && rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats_c5" -o t -- python3 "$ROOT/tools/binning_probe.py" > "$ROOT/$OUT/binning_probe_c5x32.json" 2> "$ROOT/$OUT/stats_c5.err" )
python3 tools/trace_phase_avg.py "$OUT/stats_bench/t_kernel_trace.csv" 6 18 > "$OUT/bench_kernel_trace_summary.json"   # 3 set-up + 3 warm-up launches precede the 18 timed ones
find "$OUT" -name "*kernel_stats.csv" | sort
tail -c 600 "$OUT/bench_default.json"
