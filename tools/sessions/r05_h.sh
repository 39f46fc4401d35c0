#!/bin/bash
# SQ counters of the chain kernel alone (two PMC passes over tools/bench_chain.py; no trace domains beside --kernel-trace)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
i=1
for P in "$P1" "$P2"; do
  timeout 280 rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/sq_chain_$i -o pmc -- python $R/tools/bench_chain.py 28800 > /dev/null 2>&1
  echo pass $i rc=$?
  i=$((i+1))
done
cd $R
python tools/rocpd_sq_summary.py $(find gpurun_out/sq_chain_1 gpurun_out/sq_chain_2 -name "*.db") --filter chain > gpurun_out/r05h_sq_chain.txt 2>&1
cat gpurun_out/r05h_sq_chain.txt
rm -rf gpurun_out/sq_chain_1 gpurun_out/sq_chain_2
