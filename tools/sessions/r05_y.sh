#!/bin/bash
# the backward chain after its ReLU word left the compiler's (draining) wait: parity, alone, in the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
OUT=$O/r05_chain_bwd_bits_prefetch.txt; : > $OUT
timeout 300 python -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tee -a $OUT
timeout 200 python tools/bench_chain_bwd.py 28800 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $OUT
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[1] ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3))" | tee -a $OUT
done
