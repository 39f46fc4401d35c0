#!/bin/bash
# round 4, session an: per-wave section times of the attention forward (measurement build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
( for pd in 0.1 0.0; do echo "== keep-bit dropout $pd"; PDROP=$pd timeout 120 python tools/debug/attn_fwd_sections.py 2>&1 | grep -v amdgpu.ids; done
  echo "== occupancy 2 workgroups per CU (TTSMI_ATTN_FWD_LDS=40000)"; TTSMI_ATTN_FWD_LDS=40000 timeout 120 python tools/debug/attn_fwd_sections.py 2>&1 | grep -v amdgpu.ids
  echo "== occupancy 1 workgroup per CU (TTSMI_ATTN_FWD_LDS=90000)"; TTSMI_ATTN_FWD_LDS=90000 timeout 120 python tools/debug/attn_fwd_sections.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r04an_fwd_sections.txt
cat gpurun_out/r04an_fwd_sections.txt
