#!/bin/bash
# round 4, session as: same-box A/B of the weight gradient's slab stores (float4 from transposed accumulators vs scalar)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V="TTSMI_ALLOW_LIB_OVERRIDE=1,TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_prev.so"
timeout 600 python tools/kbench.py --only wgrad --variants base "$V" 2>&1 | grep -E "^wgrad|variant" > gpurun_out/r04as_kbench.txt
cat gpurun_out/r04as_kbench.txt
: > gpurun_out/r04as_ab.txt
for i in 1 2 3; do for v in new prev; do
  if [ $v = prev ]; then export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_prev.so; else unset TTSMI_ALLOW_LIB_OVERRIDE TTSMI_LIB; fi
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] $v', 'ms_per_step', round(d['ms_per_step'], 3))" | tee -a gpurun_out/r04as_ab.txt
done; done
