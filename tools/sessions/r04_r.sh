#!/bin/bash
# round 4, session r: stage ablation of the attention forward (measurement build; TTSMI_ATTN_FWD_ABLATE bits:
# 1 no in-loop fetch, 2 no exponentials, 4 no keep-bit selects, 8 no barriers / stash, 16 / 32 no K / V fragment reads)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L="TTSMI_ALLOW_LIB_OVERRIDE=1,TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so"
timeout 600 python tools/kbench.py --only attn --variants "$L" "$L,TTSMI_ATTN_FWD_ABLATE=1" "$L,TTSMI_ATTN_FWD_ABLATE=2" \
  "$L,TTSMI_ATTN_FWD_ABLATE=4" "$L,TTSMI_ATTN_FWD_ABLATE=8" "$L,TTSMI_ATTN_FWD_ABLATE=9" "$L,TTSMI_ATTN_FWD_ABLATE=15" \
  "$L,TTSMI_ATTN_FWD_ABLATE=16" "$L,TTSMI_ATTN_FWD_ABLATE=32" "$L,TTSMI_ATTN_FWD_ABLATE=48" "$L,TTSMI_ATTN_FWD_ABLATE=63" "$L,TTSMI_ATTN_FWD_ABLATE=57" 2>&1 | grep -E "variant|fwd|^==" > gpurun_out/r04r_fwd_ablation.txt
cat gpurun_out/r04r_fwd_ablation.txt
