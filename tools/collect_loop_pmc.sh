#!/bin/bash
# Instruction, wait and LDS counters of one loop of the loop micro-benchmark (one --pmc pass per group).
# usage: bash tools/collect_loop_pmc.sh <outdir> <what> <nodes> <n>
OUT=$1; W=$2; NODES=$3; N=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/p$i -o t -- python3 $ROOT/tools/loop_bench_one.py $W $NODES $N > $ROOT/$OUT/p$i.log 2>&1
  f=$(find $ROOT/$OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 $ROOT/tools/pmc_summary.py "$f" | grep -i "loop_bench" | tail -1
done
