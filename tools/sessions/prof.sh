#!/bin/bash
# kernel-trace timeline of the current build: bash tools/sessions/prof.sh [env assignments]
# (TIMELINE_ARGS="--gaps --sequence" adds the largest main-queue gaps / the whole last step in start order)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
env "$@" timeout 280 rocprofv3 --kernel-trace -d $O/prof_tmp -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps > $O/prof_tmp.log 2>&1
echo "trace rc=$?"
python $R/tools/rocpd_timeline.py $O/prof_tmp/trace_results.db --steps 2 --top 45 $TIMELINE_ARGS > $O/prof_timeline.txt
head -8 $O/prof_timeline.txt
rm -rf $O/prof_tmp
