#!/bin/bash
# round 4, session s: attention kernels after the permlane32_swap exchange (tests + isolated timings)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -5 > gpurun_out/r04s_tests.txt
cat gpurun_out/r04s_tests.txt
timeout 600 python tools/kbench.py --only attn 2>&1 | grep -E "attn" | head -16 > gpurun_out/r04s_kbench_attn.txt
cat gpurun_out/r04s_kbench_attn.txt
