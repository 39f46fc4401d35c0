#!/bin/bash
# the chain kernels' weight streams packed on the side stream ahead of the step (TTSMI_CHAIN_PREPACK=1, default) against in line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
OUT=$O/r05_chain_prepack_ab.txt; : > $OUT
for i in 1 2 3; do for pp in 0 1; do
TTSMI_CHAIN_PREPACK=$pp python bench.py --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[1] prepack $pp ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3), 'loss', d['config'].get('loss_after'))" | tee -a $OUT
done; done
timeout 300 python -m pytest tests/test_chain_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -p no:cacheprovider -x -k "chained_blocks or b32 or B32 or parity" 2>&1 | grep -E "passed|failed" | tee -a $OUT
