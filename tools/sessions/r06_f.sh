#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_cstep_gpu.py -q -m gpu -p no:cacheprovider -k "weight_gradient or cstep or c_step or switching or outputs_of" 2>&1 | tail -6 | tee $O/r06_f_tests.txt
for w in lj-dist "configs[1]"; do
python bench.py --workload "$w" --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w ms_per_step', round(d['ms_per_step'],3), 'host', round(d['host_issue_ms_per_step'],3), 'value', round(d['value']), 'ratio', d.get('ragged_over_max_shape_per_padded_frame'))" | tee -a $O/r06_f_bench.txt
done
