#!/bin/bash
# round 4, session p: stage ablation of the weight-gradient loop and the 64-row software-pipelined variant
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 tools/probes/wgrad_probe > gpurun_out/r04p_wgrad_probe.txt 2>&1
cat gpurun_out/r04p_wgrad_probe.txt
