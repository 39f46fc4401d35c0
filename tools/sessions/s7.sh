#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 1 0; do
  TTSMI_ATTN_DROPBITS=$v timeout 280 rocprofv3 --kernel-trace -d $O/prof_s7_$v -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > $O/prof_s7_$v.log 2>&1
  echo "== DROPBITS=$v trace rc=$?"
  python $R/tools/rocpd_timeline.py $O/prof_s7_$v/trace_results.db --steps 2 --top 16
done
rm -rf $O/prof_s7_0 $O/prof_s7_1
