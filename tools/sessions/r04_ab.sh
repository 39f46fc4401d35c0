#!/bin/bash
# round 4, session ab: backward on the calling thread / one event wait per stack - tests, host issue, the ragged workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r04ab_tests.txt
cat gpurun_out/r04ab_tests.txt
: > gpurun_out/r04ab_ab.txt
for one in 0 1 0 1; do
  TTSMI_BWD_SAME_THREAD=$one timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] same_thread', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ab_ab.txt
done
for one in 0 1; do
  TTSMI_BWD_SAME_THREAD=$one timeout 600 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist same_thread', $one, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'real frames/s', round(d['value']), 'ratio', round(d['ragged_over_max_shape_per_padded_frame'], 3))" | tee -a gpurun_out/r04ab_ab.txt
done
