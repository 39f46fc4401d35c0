#!/bin/bash
# round 4, session u: stream priorities - main stream above / level with / below the weight-gradient stream
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r04u_prio.txt
for cfg in "default 0" "-1 0" "0 0" "default -1" "-1 -1" "default 1" "-1 1" "default 0"; do
  set -- $cfg
  TTSMI_WGRAD_PRIO=$2 TTSMI_PRED_PRIO=$2 timeout 300 python tools/debug/prio_ab.py $1 40 2>&1 | grep "ms/step\|Error\|error" | tail -2 >> gpurun_out/r04u_prio.txt
done
cat gpurun_out/r04u_prio.txt
