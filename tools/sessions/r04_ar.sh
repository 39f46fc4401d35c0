#!/bin/bash
# round 4, session ar: weight gradient with transposed accumulators (float4 slab stores) - tests, kbench, step, ref-default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "weight_gradient or wgrad or dense_block or conv or stack_level" 2>&1 | tail -3 > gpurun_out/r04ar_tests.txt
cat gpurun_out/r04ar_tests.txt
timeout 600 python tools/kbench.py --only wgrad 2>&1 | grep -E "^wgrad" > gpurun_out/r04ar_kbench.txt
cat gpurun_out/r04ar_kbench.txt
: > gpurun_out/r04ar_ab.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1]', 'ms_per_step', round(d['ms_per_step'], 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ar_ab.txt
done
timeout 600 python bench.py --workload ref-default --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref-default', 'ms_per_step', round(d['ms_per_step'], 3))" | tee -a gpurun_out/r04ar_ab.txt
