#!/bin/bash
# phase stamps + ablations of the chain kernel (measurement build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05f_chain_phases.txt
: > $out
export TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so
for ab in 0 1 2 8 3 11; do
  TTSMI_CHAIN_ABLATE=$ab timeout 120 python tools/probe_chain_phases.py 28800 2>&1 | grep -v amdgpu.ids >> $out
done
TTSMI_CHAIN_ABLATE=0 timeout 120 python tools/probe_chain_phases.py 6400 2>&1 | grep -v amdgpu.ids >> $out
cat $out
