#!/bin/bash
# Round 6: bounded grid of the stack keep-bit kernel (TTSMI_DROPMASK_WGS) - parity, then the step by cap
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py "tests/test_ops_gpu.py::test_keep_bit_tables_of_a_stack_in_one_launch_equal_the_single_calls" -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_bb_tests.txt
TTSMI_DROPMASK_WGS=7 timeout 300 python -m pytest "tests/test_ops_gpu.py::test_keep_bit_tables_of_a_stack_in_one_launch_equal_the_single_calls" -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tee -a $O/r06_bb_tests.txt
OUT=$O/r06_dropmask_grid_ab.txt; : > $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do for C in 0 256 512 1024; do one "TTSMI_DROPMASK_WGS=$C" "configs[1]"; done; done
for C in 0 256 512 1024 0 512; do one "TTSMI_DROPMASK_WGS=$C" lj-dist; done
