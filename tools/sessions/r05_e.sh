#!/bin/bash
# Round 5: the chain kernel, second build (every wave issues DMA pieces behind its MFMA groups, 4-stage ring, parameters staged
# in LDS): parity tests, the kernel alone, step A/B, kernel trace of the step.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
tag=${1:-r05e}
( timeout 900 python -X faulthandler -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|amdgpu.ids" | tail -30 ) > gpurun_out/${tag}_chain_tests.txt 2>&1
tail -30 gpurun_out/${tag}_chain_tests.txt
( timeout 300 python tools/bench_chain.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/${tag}_bench_chain.txt 2>&1
cat gpurun_out/${tag}_bench_chain.txt
run() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
  | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d['host_issue_ms_per_step'], 3), 'loss', d['config']['loss_after'])"; }
( for i in 1 2; do run TTSMI_DENSE_CHAIN=1; run TTSMI_DENSE_CHAIN=0; done ) > gpurun_out/${tag}_step_ab.txt 2>&1
cat gpurun_out/${tag}_step_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_chain -o chain -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --no-attention-maps > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_chain -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-150 > gpurun_out/${tag}_kernel_stats_head.txt && cat gpurun_out/${tag}_kernel_stats_head.txt
