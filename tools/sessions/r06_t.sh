#!/bin/bash
# Round 6: weight-gradient DMA ring depth 3 / 4 / 5 stages (variant libraries) - the launches alone, the wgrad parity tests, the step
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OUT=$O/r06_wgrad_ring_ab.txt; : > $OUT
L=$R/transformertts_amd/lib
for V in "" _wd4 _wd5; do
  echo "== libttsmi$V" | tee -a $OUT
  TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi$V.so timeout 300 python tools/kbench.py --only wgrad 2>&1 | grep "^wgrad" | tee -a $OUT
done
for V in _wd4 _wd5; do
  TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi$V.so timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -q -m gpu -k "wgrad" -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -5 | tee -a $OUT
done
one() {
  env TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi$1.so timeout 600 python bench.py --workload "$2" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('libttsmi$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do for V in "" _wd4 _wd5; do one "$V" "configs[1]"; done; done
for V in "" _wd4 _wd5 "" _wd4; do one "$V" lj-dist; done
