#!/bin/bash
# Round 6: precision='bf16x3' - the GEMM family, the model against the fp64 oracle, a first timing
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "bf16x3 or test_linear or test_conv1d" -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|^E " | head -30 | tee $O/r06_x3_tests.txt
timeout 900 python -m pytest tests/test_config1_parity_gpu.py -q -m gpu -k "bf16x3" -s -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|^E |config1 parity" | cut -c1-1800 | head -30 | tee -a $O/r06_x3_tests.txt
for P in bf16x3 f32; do timeout 300 python bench.py --precision $P --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$P ms_per_step', round(d['ms_per_step'],3), d['config'].get('loss_after'))" | tee -a $O/r06_x3_tests.txt; done
