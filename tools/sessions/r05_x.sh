#!/bin/bash
# Second weight-gradient lane for the encoder stack (TTSMI_WGRAD_LANES=2, the default) against one lane: headline and lj-dist,
# alternating processes on one box; then the model-level tests on the two-lane default.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OUT=$O/r05_wgrad_lanes_ab.txt
one() {  # lanes workload
  TTSMI_WGRAD_LANES=$1 python bench.py --workload $2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2 lanes $1 ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3))" | tee -a $OUT
}
for i in 1 2 3; do one 1 "configs[1]" ""; one 2 "configs[1]" ""; done


timeout 600 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py tests/test_dp_gloo.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error" | tee -a $OUT
