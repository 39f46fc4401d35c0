#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 0 3; do
  TTSMI_HGEMM_DMA=$v timeout 280 rocprofv3 --kernel-trace -d $O/prof_s3_dma$v -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > $O/prof_s3_dma$v.log 2>&1
  echo "== DMA=$v trace rc=$?"
  python $R/tools/rocpd_timeline.py $O/prof_s3_dma$v/trace_results.db --steps 2 --top 30
done
rm -rf $O/prof_s3_dma0 $O/prof_s3_dma3
