#!/bin/bash
# round 4, session y: one autograd node per StatPredictor - parity tests, host issue, the ragged workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r04y_tests.txt
cat gpurun_out/r04y_tests.txt
: > gpurun_out/r04y_ab.txt
for one in 0 1 0 1; do
  TTSMI_PRED_ONE_NODE=$one timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] one_node', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04y_ab.txt
done
for one in 0 1; do
  TTSMI_PRED_ONE_NODE=$one timeout 600 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist one_node', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'real frames/s', round(d['value']), 'ratio', round(d['ragged_over_max_shape_per_padded_frame'], 3))" | tee -a gpurun_out/r04y_ab.txt
done
