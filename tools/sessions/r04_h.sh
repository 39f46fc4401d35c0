#!/bin/bash
# round 4, session H: kernel trace of the reference-default architecture (d 384, dh 192, conv blocks) and the full -m gpu suite.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=$PWD/gpurun_out; mkdir -p $O
R=$PWD
( cd /tmp; timeout 280 rocprofv3 --kernel-trace --stats -d $O/prof_refdef -o trace -- python $R/bench.py --workload ref-default --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/r04h_refdef.json 2>/dev/null )
python tools/rocpd_kernel_stats.py $O/prof_refdef/trace_results.db $O/r04h_refdefault_kernel_stats.csv; head -45 $O/r04h_refdefault_kernel_stats.csv
python tools/rocpd_timeline.py $O/prof_refdef/trace_results.db --steps 1 --top 40 > $O/r04h_refdefault_timeline.txt 2>&1; head -60 $O/r04h_refdefault_timeline.txt
rm -rf $O/prof_refdef
cut -c1-300 $O/r04h_refdef.json
echo "== full gpu suite"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
