#!/bin/bash
# round 4, session ai: GPU timeline of the step by phase, host racing vs host out of the picture (GPU-side sleep)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/probe_phases.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04ai_phases.txt
cat gpurun_out/r04ai_phases.txt
