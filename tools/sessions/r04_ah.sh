#!/bin/bash
# round 4, session ah: kernel trace of the step with the one-pass attention backward (what do its launches take INSIDE the step?)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for f in 1 0; do
  TTSMI_ATTN_FUSED_BWD=$f timeout 280 rocprofv3 --kernel-trace --stats -d $O/prof_fused$f -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > $O/prof_fused$f.log 2>&1
  python $R/tools/rocpd_timeline.py $O/prof_fused$f/trace_results.db --steps 2 --top 14 > $O/r04ah_timeline_fused$f.txt 2>&1
  rm -rf $O/prof_fused$f
done
cd $R
head -30 $O/r04ah_timeline_fused1.txt; echo ======; head -30 $O/r04ah_timeline_fused0.txt
