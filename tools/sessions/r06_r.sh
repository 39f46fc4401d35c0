#!/bin/bash
# Round 6: hand-off events with a device-scope release (TTSMI_EVENT_SCOPE=1) / without the system fence (=2) against torch events
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
: > $O/r06_r_tests.txt
for SC in 1 2; do TTSMI_EVENT_SCOPE=$SC timeout 900 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee -a $O/r06_r_tests.txt; done
OUT=$O/r06_r_step_ab.txt; : > $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" $3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'), 'loss', d.get('loss_after'))" | tee -a $OUT
}
for i in 1 2; do for SC in 0 1 2; do one TTSMI_EVENT_SCOPE=$SC "configs[1]"; done; done
for SC in 0 1 2 0 1; do one TTSMI_EVENT_SCOPE=$SC lj-dist; done
