#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
( for SP in 1 0; do echo "== TTSMI_DENSE_CHAIN_SPLIT=$SP"; TTSMI_DENSE_CHAIN_SPLIT=$SP timeout 120 python tools/probe_chain_phases.py 6400; done ) 2>&1 | grep -v amdgpu.ids | tee $O/r06_chain_split_phases.txt
