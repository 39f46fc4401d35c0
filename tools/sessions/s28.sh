#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_config1_parity_gpu.py -q -m gpu -x -k "k256 or hgemm or config1 or parity" 2>&1 | tail -3
python tools/kbench.py --only gemm --variants TTSMI_HGEMM_K256_WIDE=0 base 2>&1 | grep -v amdgpu.ids | grep "28800\|kernel"
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run TTSMI_HGEMM_K256_WIDE=0; run A=1; done
