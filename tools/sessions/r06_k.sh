#!/bin/bash
# Round 6: the chain kernels at every row count >= 1 024 (64-row form below 16 384) inside the step: tests, headline and lj-dist A/B
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_chain_gpu.py tests/test_cstep_gpu.py tests/test_model_gpu.py tests/test_config1_parity_gpu.py tests/test_dp_gloo.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_chain64_tests.txt
OUT=$O/r06_chain64_step_ab.txt; : > $OUT
one() {
  env $1 python bench.py --workload "$2" $3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do one TTSMI_DENSE_CHAIN_MIN_ROWS=1024 "configs[1]"; one TTSMI_DENSE_CHAIN_MIN_ROWS=16384 "configs[1]"; done
one TTSMI_DENSE_CHAIN_MIN_ROWS=1024 lj-dist; one TTSMI_DENSE_CHAIN_MIN_ROWS=16384 lj-dist; one TTSMI_DENSE_CHAIN_MIN_ROWS=1024 lj-dist
