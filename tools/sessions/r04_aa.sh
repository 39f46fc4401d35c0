#!/bin/bash
# round 4, session aa: host profile with the backward pass on the calling thread
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
SINGLE=1 timeout 300 python tools/debug/host_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04aa_host_profile_single.txt
head -90 gpurun_out/r04aa_host_profile_single.txt
