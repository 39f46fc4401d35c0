#!/bin/bash
# Round 6: the 64-row chain form with loader waves at every row count (TTSMI_DENSE_CHAIN_NW=4) against the row-count rule
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
OUT=$O/r06_s_step_ab.txt; : > $OUT
( for NW in 4 8; do echo "== TTSMI_DENSE_CHAIN_NW=$NW"; TTSMI_DENSE_CHAIN_NW=$NW timeout 200 python tools/bench_chain_bwd.py 28800 20000; TTSMI_DENSE_CHAIN_NW=$NW timeout 200 python tools/bench_chain.py 28800 20000; done ) 2>&1 | grep -v amdgpu.ids | tee -a $OUT
one() {
  env $1 timeout 600 python bench.py --workload "$2" $3 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $OUT
}
for i in 1 2; do one TTSMI_DENSE_CHAIN_NW=0 "configs[1]"; one TTSMI_DENSE_CHAIN_NW=4 "configs[1]"; done
one TTSMI_DENSE_CHAIN_NW=0 lj-dist; one TTSMI_DENSE_CHAIN_NW=4 lj-dist; one TTSMI_DENSE_CHAIN_NW=0 lj-dist; one TTSMI_DENSE_CHAIN_NW=4 lj-dist
