#!/bin/bash
# round 4, session al: conv weight gradients with every tap in one launch - tests, the reference-default workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "shifted_row or conv or reference_default" 2>&1 | tail -4 > gpurun_out/r04al_tests.txt
cat gpurun_out/r04al_tests.txt
: > gpurun_out/r04al_ab.txt
for t in 0 1 0 1; do
  TTSMI_CONV_WGRAD_TAPS=$t timeout 600 python bench.py --workload ref-default --steps 15 --warmup 3 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref-default taps-in-one-launch', $t, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04al_ab.txt
done
