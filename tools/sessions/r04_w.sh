#!/bin/bash
# round 4, session w: one-pass vs two-kernel attention backward with and without the weight-gradient stream beside it
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r04w_step_ab.txt
for skip in 1 0; do for f in 0 1 0 1; do
  TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so TTSMI_DEBUG_SKIP_WGRAD=$skip TTSMI_ATTN_FUSED_BWD=$f \
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip_wgrad', $skip, 'fused_bwd', $f, 'ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3))" \
    | tee -a gpurun_out/r04w_step_ab.txt
done; done
