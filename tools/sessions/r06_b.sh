#!/bin/bash
# Round 6: the train step issued from C++ (ttsmi_ft_train_step) - parity with the per-layer path, then headline and lj-dist A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_cstep_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -30 | tee $O/r06_cstep_parity.txt
OUT=$O/r06_cstep_ab.txt
one() {
  env $1 python bench.py --workload "$2" --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>$O/r06_b_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 $2 ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3), 'value', round(d['value']))" | tee -a $OUT
  tail -3 $O/r06_b_err.txt
}
one TTSMI_CSTEP=1 "configs[1]"; one TTSMI_CSTEP=0 "configs[1]"; one TTSMI_CSTEP=1 "lj-dist"; one TTSMI_CSTEP=0 "lj-dist"; one TTSMI_CSTEP=1 "configs[1]"; one TTSMI_CSTEP=1 "lj-dist"
timeout 900 python -m pytest tests/test_dp_gloo.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 | tee $O/r06_cstep_model_tests.txt
