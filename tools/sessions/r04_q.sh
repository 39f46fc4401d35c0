#!/bin/bash
# round 4, session q: VALU issue rates (tools/probes/valu_rate_probe.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 tools/probes/valu_rate_probe > gpurun_out/r04q_valu_rates.txt 2>&1
cat gpurun_out/r04q_valu_rates.txt
