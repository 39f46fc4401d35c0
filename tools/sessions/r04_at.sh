#!/bin/bash
# round 4, session at: same-box A/B of the dh = 64 backward kernels compiled for one more workgroup per CU
# (dK/dV: 3 instead of 2 - 168 registers + 41 dwords of scratch; dQ: 4 instead of 3 - 128 registers + 53 dwords)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=$PWD/transformertts_amd/lib
V=""
for n in dkv3 dq4 dkv3dq4; do V="$V TTSMI_ALLOW_LIB_OVERRIDE=1,TTSMI_LIB=$L/libttsmi_$n.so"; done
timeout 600 python tools/kbench.py --only attn --variants base $V 2>&1 | grep -E "^attn|variant" > gpurun_out/r04at_kbench.txt
cat gpurun_out/r04at_kbench.txt
: > gpurun_out/r04at_ab.txt
for i in 1 2; do for v in base dkv3 dq4 dkv3dq4; do
  if [ $v = base ]; then unset TTSMI_ALLOW_LIB_OVERRIDE TTSMI_LIB; else export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi_$v.so; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] $v', 'ms_per_step', round(d['ms_per_step'], 3), 'loss', d['config'].get('loss_after'))" | tee -a gpurun_out/r04at_ab.txt
done; done
