#!/bin/bash
# HBM traffic of every kernel from the TCC counters, collected exactly as MI355X_MICROARCH.md (HBM / rocprofv3
# section) prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes, with --kernel-trace only.
# Run on the GPU box from the repo root:
#   bash tools/collect_traffic.sh <outdir> [bench.py args...]            (the bench batch)
#   PROBE=1 bash tools/collect_traffic.sh <outdir>                        (tools/binning_probe.py: 32 dense C5 frames)
set -e
OUT=${1:-gpurun_out/traffic}; shift || true
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  if [ -n "$PROBE" ]; then
    MAX_ITER=1 REPS=3 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/$OUT/$C" -o t -- python3 "$ROOT/tools/binning_probe.py" > "$ROOT/$OUT/$C.probe.json" 2> "$ROOT/$OUT/$C.err"
  else
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/$OUT/$C" -o t -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --regions 1 --no-cpu-baseline --no-host-path "$@" > "$ROOT/$OUT/$C.bench.json" 2> "$ROOT/$OUT/$C.err"
  fi
done
python3 "$ROOT/tools/traffic_summary.py" "$ROOT/$OUT" > "$ROOT/$OUT/traffic.json"
cat "$ROOT/$OUT/traffic.json"
