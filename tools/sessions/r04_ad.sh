#!/bin/bash
# round 4, session ad: full-row GEMM + LN with two 256-thread workgroups per CU (TTSMI_ROWGEMM_DMA2)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tools/kbench.py --only rowgemm --variants TTSMI_ROWGEMM_DMA2=0 TTSMI_ROWGEMM_DMA2=1 2>&1 | grep -E "rowg|variant" > gpurun_out/r04ad_kbench.txt
cat gpurun_out/r04ad_kbench.txt
TTSMI_ROWGEMM_DMA2=1 timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "rowgemm or full_row or hgemm_ln" 2>&1 | tail -8 > gpurun_out/r04ad_tests.txt
cat gpurun_out/r04ad_tests.txt
: > gpurun_out/r04ad_ab.txt
for one in 0 1 0 1; do
  TTSMI_ROWGEMM_DMA2=$one timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] dma2', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ad_ab.txt
done
