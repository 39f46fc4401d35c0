#!/bin/bash
# chain kernel build N: parity tests, phase stamps (measurement build), kernel alone, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r05g}
( timeout 900 python -X faulthandler -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -30 ) > gpurun_out/${tag}_chain_tests.txt 2>&1
tail -12 gpurun_out/${tag}_chain_tests.txt
( for ab in 0 1; do TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so TTSMI_CHAIN_ABLATE=$ab timeout 120 python tools/probe_chain_phases.py 28800 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/${tag}_chain_phases.txt 2>&1
cat gpurun_out/${tag}_chain_phases.txt
( timeout 300 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${tag}_bench_chain.txt 2>&1
cat gpurun_out/${tag}_bench_chain.txt
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
  | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d['host_issue_ms_per_step'], 3), 'loss', d['config']['loss_after'])"; }
( for i in 1 2; do run TTSMI_DENSE_CHAIN=1; run TTSMI_DENSE_CHAIN=0; done ) > gpurun_out/${tag}_step_ab.txt 2>&1
cat gpurun_out/${tag}_step_ab.txt
