#!/bin/bash
# the exact-fp32 step's kernel breakdown (where would a 3 x bf16 product family pay?)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_f32 -o trace -- python $R/bench.py --precision f32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_f32.log 2>&1
python $R/tools/rocpd_kernel_stats.py $O/prof_f32/trace_results.db $O/r06_f32_kernel_stats.csv
python $R/tools/rocpd_timeline.py $O/prof_f32/trace_results.db --steps 2 --top 40 > $O/r06_timeline_f32.txt 2>&1
rm -rf $O/prof_f32
head -30 $O/r06_f32_kernel_stats.csv | cut -c1-150
