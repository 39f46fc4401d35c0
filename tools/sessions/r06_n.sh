#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
( for M in 6400 12000; do python tools/probe_chain_phases.py $M; TTSMI_DENSE_CHAIN_NW=8 python tools/probe_chain_phases.py $M; done ) 2>&1 | grep -v amdgpu.ids | tee $O/r06_chain64_phases.txt
