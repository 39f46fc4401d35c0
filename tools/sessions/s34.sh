#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 280 rocprofv3 --kernel-trace -d $O/prof_tmp -o trace -- python $R/bench.py --workload ref-default --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps > $O/prof_tmp.log 2>&1
python $R/tools/rocpd_timeline.py $O/prof_tmp/trace_results.db --steps 2 --top 30
rm -rf $O/prof_tmp
