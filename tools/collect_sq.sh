#!/bin/bash
# Instruction-issue counters of every kernel (one --pmc pass per group, --kernel-trace only).
# Run on the GPU box from the repo root:  bash tools/collect_sq.sh <outdir> <bench args...>
set -e
OUT=${1:-gpurun_out/sq}; shift || true
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$ROOT/$OUT/p$i" -o t -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --streams 1 --regions 1 --no-cpu-baseline --no-host-path --no-profile-events "$@" > "$ROOT/$OUT/p$i.bench.json" 2> "$ROOT/$OUT/p$i.err"
  f=$(find "$ROOT/$OUT/p$i" -name "*counter_collection.csv" | head -1)
  python3 "$ROOT/tools/pmc_summary.py" "$f" | tee -a "$ROOT/$OUT/summary.txt"
done
