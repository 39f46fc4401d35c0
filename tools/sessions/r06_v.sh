#!/bin/bash
# Round 6: two hand-offs per chained block (W2 behind the backward chain) A/B; predict latency by chain threshold
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
TTSMI_WGRAD_EVENTS=3 timeout 600 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_v_tests.txt
OUT=$O/r06_two_handoffs_ab.txt; : > $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" $3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2 3; do one TTSMI_WGRAD_EVENTS=4 "configs[1]"; one TTSMI_WGRAD_EVENTS=3 "configs[1]"; done
one TTSMI_WGRAD_EVENTS=4 lj-dist; one TTSMI_WGRAD_EVENTS=3 lj-dist; one TTSMI_WGRAD_EVENTS=4 lj-dist; one TTSMI_WGRAD_EVENTS=3 lj-dist
for T in 1024 4096 1024 4096; do TTSMI_DENSE_CHAIN_MIN_ROWS=$T timeout 300 python bench.py --workload predict 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CHAIN_MIN_ROWS=$T predict', {k:v for k,v in d.items() if 'p50' in k or 'p90' in k or k=='value'}, json.dumps(d)[:300])" | tee -a $O/r06_predict_chain_rows.txt; done
