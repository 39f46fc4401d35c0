#!/bin/bash
# round 4, session ap: the forward's next-tile loads issued behind block 0's S product - tests, section times, kbench, the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "attn or attention" 2>&1 | tail -3 > gpurun_out/r04ap_tests.txt
cat gpurun_out/r04ap_tests.txt
( export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
  for pd in 0.1; do echo "== keep-bit dropout $pd"; PDROP=$pd timeout 120 python tools/debug/attn_fwd_sections.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r04ap_fwd_sections.txt
cat gpurun_out/r04ap_fwd_sections.txt
timeout 600 python tools/kbench.py --only attn 2>&1 | grep -E "attn   fwd" | head -6 > gpurun_out/r04ap_kbench.txt
cat gpurun_out/r04ap_kbench.txt
: > gpurun_out/r04ap_ab.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1]', 'ms_per_step', round(d['ms_per_step'], 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ap_ab.txt
done
