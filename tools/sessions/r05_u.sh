#!/bin/bash
# the backward chain: kernel parity, model-level tests, step A/B (forward + backward chain / forward chain only / no chain)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r05u}
( timeout 900 python -X faulthandler -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -40 ) > gpurun_out/${tag}_chain_tests.txt 2>&1
tail -40 gpurun_out/${tag}_chain_tests.txt
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
  | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d['host_issue_ms_per_step'], 3), 'loss', d['config']['loss_after'])"; }
( for i in 1 2; do run A=1; run TTSMI_DENSE_CHAIN_BWD=0; run TTSMI_DENSE_CHAIN=0; done ) > gpurun_out/${tag}_step_ab.txt 2>&1
cat gpurun_out/${tag}_step_ab.txt
( timeout 1200 python -X faulthandler -m pytest tests/test_config1_parity_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -25 ) > gpurun_out/${tag}_model_tests.txt 2>&1
tail -8 gpurun_out/${tag}_model_tests.txt
