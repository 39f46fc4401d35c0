#!/bin/bash
# round 4, session av: the full-row GEMM + LN kernel with its waves split by role (fillers / multipliers): parity tests of
# the fused kernels with the knob on, then the same-box kbench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TTSMI_ROWGEMM_PS=1 timeout 300 python -m pytest tests/test_bench_shapes_gpu.py -q -m gpu -p no:cacheprovider -x -k "fused_gemm_layernorm or fused_dgrad_layernorm" 2>&1 | tail -5 | tee gpurun_out/r04av_tests.txt
timeout 300 python tools/kbench.py --only rowgemm --variants base TTSMI_ROWGEMM_PS=1 2>&1 | grep -E "^rowg|variant" | tee gpurun_out/r04av_kbench.txt
