#!/bin/bash
# Round 5: first run of the row-local chain kernel - its parity tests, the kernel alone against the four launches it replaces,
# the model-level tests that go through it, and the step with / without it (each variant its own process).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -X faulthandler -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -40 ) > gpurun_out/r05d_chain_tests.txt 2>&1
tail -40 gpurun_out/r05d_chain_tests.txt
( timeout 300 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r05d_bench_chain.txt 2>&1
cat gpurun_out/r05d_bench_chain.txt
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
  | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d['host_issue_ms_per_step'], 3), 'loss', d['config']['loss_after'])"; }
( for i in 1 2; do run TTSMI_DENSE_CHAIN=1; run TTSMI_DENSE_CHAIN=0; done ) > gpurun_out/r05d_step_ab.txt 2>&1
cat gpurun_out/r05d_step_ab.txt
( timeout 1200 python -X faulthandler -m pytest tests/test_config1_parity_gpu.py tests/test_model_gpu.py tests/test_bench_shapes_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -30 ) > gpurun_out/r05d_model_tests.txt 2>&1
tail -30 gpurun_out/r05d_model_tests.txt
