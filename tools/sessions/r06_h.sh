#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cstep_gpu.py tests/test_dp_gloo.py -q -m gpu -p no:cacheprovider 2>&1 | tail -5 | tee $O/r06_h_tests.txt
cd /tmp
timeout 280 rocprofv3 --kernel-trace -d $O/prof_hd -o trace -- python $R/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_hd.log 2>&1
python $R/tools/rocpd_timeline.py $O/prof_hd/trace_results.db --steps 2 --top 50 --gaps > $O/r06_timeline_bf16.txt 2>&1; head -22 $O/r06_timeline_bf16.txt
rm -rf $O/prof_hd
cd $R
for i in 1 2; do python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(d['ms_per_step'],3), 'host burst', round(d['host_issue_burst_ms_per_step'],3))"; done | tee $O/r06_h_bench.txt
