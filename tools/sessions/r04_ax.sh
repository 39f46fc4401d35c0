#!/bin/bash
# round 4, session ax: gradients of multiply-used tensors summed in place on the per-layer path (ops.GradSink): parity
# tests of the paths that use it, then the reference-default and exact-fp32 steps with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_zz_reference_source_gpu.py tests/test_ops_gpu.py tests/test_config1_parity_gpu.py tests/test_dp_gloo.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee gpurun_out/r04ax_tests.txt
: > gpurun_out/r04ax_ab.txt
for i in 1 2; do for v in 1 0; do
  TTSMI_GRAD_SINK=$v timeout 300 python bench.py --workload ref-default --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref-default sink $v', 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ax_ab.txt
done; done
for v in 1 0; do
  TTSMI_GRAD_SINK=$v timeout 300 python bench.py --precision f32 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 sink $v', 'ms_per_step', round(d['ms_per_step'], 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ax_ab.txt
done
