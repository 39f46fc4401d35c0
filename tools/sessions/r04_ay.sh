#!/bin/bash
# round 4, session ay: FFT butterflies / complex products on packed fp32 ops with per-lane negation and operand selects
# (inline asm): mel + Griffin-Lim parity tests, then the mel workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_golden.py tests/test_griffinlim.py -q -m gpu -p no:cacheprovider -x -k "mel or audio or griffin or istft or reconstruct or nnls or fixtures" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r04ay_tests.txt
for i in 1 2; do
timeout 300 python bench.py --workload mel --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mel GB/s', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3), 'frac', d['roofline']['frac'])" | tee -a gpurun_out/r04ay_mel.txt
done
