#!/bin/bash
# round 4, session af: 256 x 256-tile GEMM, DMA pieces spread over the k-step vs issued in a burst behind the barrier
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "256_tile" 2>&1 | tail -3 > gpurun_out/r04af_tests.txt
cat gpurun_out/r04af_tests.txt
: > gpurun_out/r04af_ab.txt
for b in 1 0 1 0; do
  echo "TTSMI_HGEMM_T256_BURST=$b" >> gpurun_out/r04af_ab.txt
  TTSMI_HGEMM_T256_BURST=$b timeout 300 python tools/debug/conv_gemm_ab.py 2>&1 | grep "dma256" >> gpurun_out/r04af_ab.txt
done
cat gpurun_out/r04af_ab.txt
