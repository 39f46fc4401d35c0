#!/bin/bash
# Round 6: batched pack / keep-bit launches in the C step, and ONE hand-off per chained block (TTSMI_WGRAD_EVENTS=3) A/B
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py "tests/test_ops_gpu.py::test_keep_bit_tables_of_a_stack_in_one_launch_equal_the_single_calls" -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_q_tests.txt
grep -q failed $O/r06_q_tests.txt && exit 1
TTSMI_WGRAD_EVENTS=3 timeout 600 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee -a $O/r06_q_tests.txt
OUT=$O/r06_q_step_ab.txt; : > $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" $3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'), 'burst', d.get('host_issue_burst_ms_per_step'))" | tee -a $OUT
}
for i in 1 2; do one TTSMI_WGRAD_EVENTS=4 "configs[1]"; one TTSMI_WGRAD_EVENTS=3 "configs[1]"; done
one TTSMI_WGRAD_EVENTS=4 lj-dist; one TTSMI_WGRAD_EVENTS=3 lj-dist; one TTSMI_WGRAD_EVENTS=4 lj-dist; one TTSMI_WGRAD_EVENTS=3 lj-dist
