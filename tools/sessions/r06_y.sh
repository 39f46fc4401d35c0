#!/bin/bash
# Round 6: the SPLIT forward chain form (two workgroups per 64-row tile) - parity (timeout-wrapped: a handshake that hangs must not
# take the box), alone, in the step
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_chain_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert|Timeout" | head -10 | tee $O/r06_y_tests.txt
grep -q "passed" $O/r06_y_tests.txt || exit 1
grep -q failed $O/r06_y_tests.txt && exit 1
timeout 600 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee -a $O/r06_y_tests.txt
OUT=$O/r06_chain_split_ab.txt; : > $OUT
for SP in 1 0; do
  echo "== TTSMI_DENSE_CHAIN_SPLIT=$SP" | tee -a $OUT
  TTSMI_DENSE_CHAIN_SPLIT=$SP timeout 200 python tools/bench_chain.py 6400 2500 8192 2>&1 | grep -v amdgpu.ids | tee -a $OUT; TTSMI_DENSE_CHAIN_SPLIT=$SP timeout 200 python tools/bench_chain_bwd.py 6400 8192 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
one() {
  env $1 timeout 600 python bench.py --workload "$2" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do one "TTSMI_DENSE_CHAIN_SPLIT=1" "configs[1]"; one "TTSMI_DENSE_CHAIN_SPLIT=0" "configs[1]"; done
one "TTSMI_DENSE_CHAIN_SPLIT=1" lj-dist; one "TTSMI_DENSE_CHAIN_SPLIT=0" lj-dist; one "TTSMI_DENSE_CHAIN_SPLIT=1" lj-dist; one "TTSMI_DENSE_CHAIN_SPLIT=0" lj-dist
