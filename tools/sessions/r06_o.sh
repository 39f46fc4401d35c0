#!/bin/bash
# Round 6: loader waves beside the 64-row chain forms - parity first, then the launch alone (phase stamps) and the step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_loaders_tests.txt
grep -q failed $O/r06_loaders_tests.txt && exit 1
( export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
  for M in 6400 12000; do for L in 1 0; do echo "== TTSMI_DENSE_CHAIN_LOADERS=$L"; TTSMI_DENSE_CHAIN_LOADERS=$L timeout 120 python tools/probe_chain_phases.py $M; done; done ) 2>&1 | grep -v amdgpu.ids | tee $O/r06_loaders_phases.txt
OUT=$O/r06_loaders_step_ab.txt; : > $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" $3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do one TTSMI_DENSE_CHAIN_LOADERS=1 "configs[1]"; one TTSMI_DENSE_CHAIN_LOADERS=0 "configs[1]"; done
one TTSMI_DENSE_CHAIN_LOADERS=1 lj-dist; one TTSMI_DENSE_CHAIN_LOADERS=0 lj-dist; one TTSMI_DENSE_CHAIN_LOADERS=1 lj-dist
