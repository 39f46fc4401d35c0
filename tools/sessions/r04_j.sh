#!/bin/bash
# round 4, session J: the stack-level launcher and the dual-X Wo weight gradient - tests, step A/Bs, lj-dist.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py tests/test_training_curve_gpu.py -x -q -m gpu 2>&1 | tail -6
run() { env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['host_issue_ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run A=1; run TTSMI_DENSE_STACK=0; run TTSMI_WGRAD_WO_DUAL=0; done
echo "== lj-dist"
for e in A=1 TTSMI_DENSE_STACK=0; do env $e timeout 200 python bench.py --workload lj-dist --steps 150 --warmup 20 --lj-samples 2048 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$e', {k: d[k] for k in ('value', 'ms_per_step','host_stall_ms_per_step','host_issue_ms_per_step','distinct_batch_shapes')})"; done
python -c "
import json; d=json.load(open('gpurun_out/bf16_vs_f32_curve.json')); print({k:v for k,v in d.items() if 'curve' not in k})"
