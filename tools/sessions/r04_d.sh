#!/bin/bash
# round 4, session D: which case of the one-pass backward test dies; the chain kernel with its round trips hidden (kbench,
# step A/B); lj-dist with and without the producer thread + a host profile sorted by own time; the training curve.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== one-pass backward, case by case"
timeout 400 python tools/debug/fused_bwd_cases.py all 2>&1 | cut -c1-400 | tee $O/r04d_fused_cases.txt
echo "== kbench attn"
timeout 300 python tools/kbench.py --only attn 2>&1 | grep -v amdgpu.ids | grep "900    64\|200    64\|kernel" | tee $O/r04d_kbench_attn.txt
echo "== step A/B"
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run TTSMI_ATTN_FUSED_BWD=0; run TTSMI_ATTN_FUSED_BWD=1; done
echo "== training curve"
timeout 300 python -m pytest tests/test_training_curve_gpu.py -q -m gpu 2>&1 | tail -3
python -c "
import json; d=json.load(open('gpurun_out/bf16_vs_f32_curve.json')); print({k:v for k,v in d.items() if 'curve' not in k})"
echo "== lj-dist: producer thread vs preloaded batches"
for a in "" "--lj-preload"; do timeout 200 python bench.py --workload lj-dist --steps 60 --warmup 15 --lj-samples 2048 $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$a', {k: d[k] for k in ('ms_per_step','host_stall_ms_per_step','host_issue_ms_per_step','distinct_batch_shapes','ragged_over_max_shape_per_padded_frame')}, d['max_shape'])"; done
echo "== lj-dist: host profile (preloaded), by own time"
timeout 280 python -m cProfile -o /tmp/lj.prof bench.py --workload lj-dist --steps 40 --warmup 10 --lj-samples 1024 --lj-preload > /dev/null 2>&1
python -c "
import pstats; p=pstats.Stats('/tmp/lj.prof'); p.sort_stats('tottime').print_stats(40)" 2>&1 | tail -50 | cut -c1-200 | tee $O/r04d_lj_host_profile.txt
