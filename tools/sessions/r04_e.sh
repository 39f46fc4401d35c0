#!/bin/bash
# round 4, session E: the chain kernel with LDS-only barriers (kbench, step A/B), lj-dist through the pinned staging ring,
# the training curve with its chaos floor, the fused tests under pytest.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== kbench attn"
timeout 300 python tools/kbench.py --only attn 2>&1 | grep -v amdgpu.ids | grep "900    64\|200    64\|kernel" | tee $O/r04e_kbench_attn.txt
echo "== step A/B"
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run TTSMI_ATTN_FUSED_BWD=0; run TTSMI_ATTN_FUSED_BWD=1; done
echo "== lj-dist: producer thread (pinned ring) vs preloaded batches"
for a in "" "--lj-preload"; do timeout 200 python bench.py --workload lj-dist --steps 100 --warmup 15 --lj-samples 2048 $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$a', {k: d[k] for k in ('value', 'ms_per_step','host_stall_ms_per_step','host_issue_ms_per_step','distinct_batch_shapes','ragged_over_max_shape_per_padded_frame')}, d['max_shape'])"; done
echo "== training curve"
timeout 300 python -m pytest tests/test_training_curve_gpu.py -q -m gpu 2>&1 | tail -3
python -c "
import json; d=json.load(open('gpurun_out/bf16_vs_f32_curve.json')); print({k:v for k,v in d.items() if 'curve' not in k})"
echo "== fused tests under pytest"
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -x -q -m gpu -k "one_pass or keep_bit" > $O/r04e_fused_tests.txt 2>&1; echo rc=$?; head -5 $O/r04e_fused_tests.txt | cut -c1-200; tail -5 $O/r04e_fused_tests.txt | cut -c1-300
