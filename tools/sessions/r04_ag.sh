#!/bin/bash
# round 4, session ag: full-row GEMM + LN, DMA pieces spread over the k-step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "fused_gemm_layernorm or fused_dgrad" 2>&1 | tail -3 > gpurun_out/r04ag_tests.txt
cat gpurun_out/r04ag_tests.txt
timeout 600 python tools/kbench.py --only rowgemm --variants TTSMI_ROWGEMM_SPREAD=0 TTSMI_ROWGEMM_SPREAD=1 2>&1 | grep -E "fused" > gpurun_out/r04ag_kbench.txt
cat gpurun_out/r04ag_kbench.txt
: > gpurun_out/r04ag_ab.txt
for one in 0 1 0 1; do
  TTSMI_ROWGEMM_SPREAD=$one timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] spread', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ag_ab.txt
done
