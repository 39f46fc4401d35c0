#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "keep_bit_attention" 2>&1 | grep -E "assert|Error|worst|passed|failed" | head -20 > gpurun_out/r04t_tests.txt
cat gpurun_out/r04t_tests.txt
