#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
( timeout 120 python tools/probe_chain_phases.py 28800; timeout 120 python tools/probe_chain_phases.py 6400 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r05s_chain16_phases.txt
cat gpurun_out/r05s_chain16_phases.txt
