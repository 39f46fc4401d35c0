#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bf16x3" -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -20 | tee $O/r06_x3_tests.txt
timeout 900 python -m pytest tests/test_config1_parity_gpu.py -q -m gpu -k "bf16x3" -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|^E " | head -10 | tee -a $O/r06_x3_tests.txt
for P in bf16x3 f32; do timeout 300 python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P ms_per_step', round(d['ms_per_step'],3), d['config'].get('loss_after'))" | tee -a $O/r06_x3_tests.txt; done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_x3 -o trace -- python $R/bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also > $O/prof_x3.log 2>&1
python $R/tools/rocpd_kernel_stats.py $O/prof_x3/trace_results.db $O/r06_x3_kernel_stats.csv
rm -rf $O/prof_x3
head -12 $O/r06_x3_kernel_stats.csv | cut -c1-150
