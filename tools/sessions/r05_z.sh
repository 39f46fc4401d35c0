#!/bin/bash
# epilogue constants of the full-row GEMM + LN kernels and the forward chain's step counter requested in front of the first DMA
# issue: parity of everything that launches them, then the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
OUT=$O/r05_early_epilogue_constants.txt; : > $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bench_shapes_gpu.py tests/test_chain_gpu.py tests/test_model_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -p no:cacheprovider -x -n 4 2>&1 | grep -E "passed|failed|error" | tee -a $OUT
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[1] ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3))" | tee -a $OUT
done
