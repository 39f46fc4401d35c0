#!/bin/bash
# SQ counter passes (8 slots each) over two bench steps; summarised per kernel by tools/rocpd_sq_summary.py
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline $*"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2"; do
  timeout 280 rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/sq_${TAG}_$i -o pmc -- python $R/bench.py $ARGS > /dev/null 2>&1
  echo pass $i rc=$?
  i=$((i+1))
done
