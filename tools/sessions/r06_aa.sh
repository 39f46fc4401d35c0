#!/bin/bash
# Round 6: the split forms' XCD-aware exchange - parity, two-stream stress (timeout-wrapped), alone, phases, the step
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|Timeout" | head -10 | tee $O/r06_aa_tests.txt
grep -q "passed" $O/r06_aa_tests.txt || exit 1
grep -q failed $O/r06_aa_tests.txt && exit 1
( timeout 120 python tools/probe_chain_split_two_streams.py 6400 50; echo "rc=$?"; timeout 120 python tools/probe_chain_split_two_streams.py 2500 50; echo "rc=$?" ) 2>&1 | grep -v amdgpu.ids | tee $O/r06_chain_split_two_streams.txt
timeout 600 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee -a $O/r06_aa_tests.txt
OUT=$O/r06_chain_split_xcd_ab.txt; : > $OUT
( timeout 200 python tools/bench_chain.py 6400 2500 8192; timeout 200 python tools/bench_chain_bwd.py 6400 8192 ) 2>&1 | grep -v amdgpu.ids | tee -a $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do one "TTSMI_DENSE_CHAIN_SPLIT=1" "configs[1]"; one "TTSMI_DENSE_CHAIN_SPLIT=0" "configs[1]"; done
one "TTSMI_DENSE_CHAIN_SPLIT=1" lj-dist; one "TTSMI_DENSE_CHAIN_SPLIT=0" lj-dist; one "TTSMI_DENSE_CHAIN_SPLIT=1" lj-dist
