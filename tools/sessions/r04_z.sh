#!/bin/bash
# round 4, session z: host time inside the C-ABI calls vs around them
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/debug/host_split.py 30 2>&1 | grep -v amdgpu.ids > gpurun_out/r04z_host_split.txt
timeout 300 python tools/debug/host_split.py 30 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04z_host_split.txt
cat gpurun_out/r04z_host_split.txt
