#!/bin/bash
# Round 6: A/B of the working-tree library against the previous commit's (lib/libttsmi_prev.so): cstep parity, headline, lj-dist
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cstep_gpu.py tests/test_config1_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_x_tests.txt
OUT=$O/r06_x_ab.txt; : > $OUT
L=$R/transformertts_amd/lib
one() {
  env TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi$1.so timeout 600 python bench.py --workload "$2" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('libttsmi$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2 3; do one "" "configs[1]"; one _prev "configs[1]"; done
for V in "" _prev "" _prev; do one "$V" lj-dist; done
