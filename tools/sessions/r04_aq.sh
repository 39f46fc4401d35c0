#!/bin/bash
# round 4, session aq: same-box A/B - the forward's next-tile loads spread behind the four MFMA groups vs all behind block 0's S
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V="TTSMI_ALLOW_LIB_OVERRIDE=1,TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_nospread.so"
timeout 600 python tools/kbench.py --only attn --variants base "$V" base "$V" 2>&1 | grep -E "attn   fwd" | head -5 > gpurun_out/r04aq_kbench.txt
cat gpurun_out/r04aq_kbench.txt
: > gpurun_out/r04aq_ab.txt
for i in 1 2 3; do for v in spread nospread; do
  if [ $v = nospread ]; then export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_nospread.so; else unset TTSMI_ALLOW_LIB_OVERRIDE TTSMI_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] $v', 'ms_per_step', round(d['ms_per_step'], 3))" | tee -a gpurun_out/r04aq_ab.txt
done; done
