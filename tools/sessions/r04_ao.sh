#!/bin/bash
# round 4, session ao: 16-byte tile staging in the attention kernels - forward section times, the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
  for pd in 0.1 0.0; do echo "== keep-bit dropout $pd"; PDROP=$pd timeout 120 python tools/debug/attn_fwd_sections.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r04ao_fwd_sections.txt
cat gpurun_out/r04ao_fwd_sections.txt
: > gpurun_out/r04ao_ab.txt
for i in 1 2 3; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1]', 'ms_per_step', round(d['ms_per_step'], 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04ao_ab.txt
done
