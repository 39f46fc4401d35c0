#!/bin/bash
# round 4, session x: host cost of a launch (raw / wrapper / torch / autograd)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/debug/launch_cost.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04x_launch_cost.txt
cat gpurun_out/r04x_launch_cost.txt
