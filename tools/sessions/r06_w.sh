#!/bin/bash
# Round 6: chain forms with FOUR loader waves - two compute waves (32-row tiles) and four compute waves - parity, alone, in the step
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
: > $O/r06_w_tests.txt
TTSMI_DENSE_CHAIN_NW=2 timeout 900 python -m pytest tests/test_chain_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee -a $O/r06_w_tests.txt
TTSMI_DENSE_CHAIN_LOADERS=4 timeout 900 python -m pytest tests/test_chain_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee -a $O/r06_w_tests.txt
grep -q failed $O/r06_w_tests.txt && exit 1
OUT=$O/r06_chain_loaders4_ab.txt; : > $OUT
for CFG in "TTSMI_DENSE_CHAIN_NW=0 TTSMI_DENSE_CHAIN_LOADERS=2" "TTSMI_DENSE_CHAIN_NW=0 TTSMI_DENSE_CHAIN_LOADERS=4" "TTSMI_DENSE_CHAIN_NW=2 TTSMI_DENSE_CHAIN_LOADERS=2"; do
  echo "== $CFG" | tee -a $OUT
  ( env $CFG timeout 200 python tools/bench_chain.py 6400 2500 12000; env $CFG timeout 200 python tools/bench_chain_bwd.py 6400 2500 12000 ) 2>&1 | grep -v amdgpu.ids | tee -a $OUT
done
one() {
  env $1 timeout 600 python bench.py --workload "$2" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do one "TTSMI_DENSE_CHAIN_LOADERS=2" "configs[1]"; one "TTSMI_DENSE_CHAIN_LOADERS=4" "configs[1]"; done
one "TTSMI_DENSE_CHAIN_LOADERS=2" lj-dist; one "TTSMI_DENSE_CHAIN_LOADERS=4" lj-dist; one "TTSMI_DENSE_CHAIN_LOADERS=2" lj-dist; one "TTSMI_DENSE_CHAIN_LOADERS=4" lj-dist
