#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o trace -- python $R/bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_x3.log 2>&1
python $R/tools/rocpd_kernel_stats.py $O/prof_x3/trace_results.db $O/r06_x3_kernel_stats.csv
python $R/tools/rocpd_timeline.py $O/prof_x3/trace_results.db --steps 2 --top 30 > $O/r06_timeline_x3.txt 2>&1
rm -rf $O/prof_x3
head -14 $O/r06_x3_kernel_stats.csv | cut -c1-150; head -12 $O/r06_timeline_x3.txt
