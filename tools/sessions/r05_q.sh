#!/bin/bash
# the chain kernel's second form (eight 16-row waves, two per SIMD): parity tests in both forms, alone against the first form and
# the four launches, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r05q}
( for f in 16 32; do TTSMI_DENSE_CHAIN_FORM=$f timeout 900 python -X faulthandler -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -25; done ) > gpurun_out/${tag}_chain_tests.txt 2>&1
tail -30 gpurun_out/${tag}_chain_tests.txt
( for f in 16 32; do echo "== TTSMI_DENSE_CHAIN_FORM=$f"; TTSMI_DENSE_CHAIN_FORM=$f timeout 300 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/${tag}_bench_chain.txt 2>&1
cat gpurun_out/${tag}_bench_chain.txt
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
  | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d['host_issue_ms_per_step'], 3), 'loss', d['config']['loss_after'])"; }
( for i in 1 2; do run TTSMI_DENSE_CHAIN=1 TTSMI_DENSE_CHAIN_FORM=16; run TTSMI_DENSE_CHAIN=0; run TTSMI_DENSE_CHAIN=1 TTSMI_DENSE_CHAIN_FORM=32; done ) > gpurun_out/${tag}_step_ab.txt 2>&1
cat gpurun_out/${tag}_step_ab.txt
