#!/bin/bash
# round 4, session am: kernel trace of the reference-default step (after the 256-tile GEMM and the one-launch conv taps)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp; timeout 280 rocprofv3 --kernel-trace --stats -d $O/prof_refdef2 -o trace -- python $R/bench.py --workload ref-default --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > /dev/null 2>&1 )
python tools/rocpd_kernel_stats.py $O/prof_refdef2/trace_results.db $O/r04am_refdefault_kernel_stats.csv
python tools/rocpd_timeline.py $O/prof_refdef2/trace_results.db --steps 1 --top 45 --gaps > $O/r04am_refdefault_timeline.txt 2>&1
rm -rf $O/prof_refdef2
cat $O/r04am_refdefault_timeline.txt
