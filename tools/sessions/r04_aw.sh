#!/bin/bash
# round 4, session aw: stream objects remembered while pinned (no torch.cuda.current_stream() per event / side-stream
# context): model + dp + dataset tests, then host issue on configs[1] and the ragged workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gloo.py tests/test_datasets.py tests/test_config1_parity_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4 | tee gpurun_out/r04aw_tests.txt
: > gpurun_out/r04aw_host.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04aw_host.txt
  timeout 600 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'real frames/s', round(d['value']), 'ratio', round(d['ragged_over_max_shape_per_padded_frame'], 3))" | tee -a gpurun_out/r04aw_host.txt
done
