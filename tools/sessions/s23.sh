#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -x 2>&1 | tail -4
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run A=1; run TTSMI_HGEMM_K256_MASK=0; run TTSMI_DENSE_SPLIT_DGRAD=0; done
