#!/bin/bash
# round 4, session o: what the memory pipeline gives the weight gradient's stream (tools/probes/stream_tile_probe.hip);
# the step with the weight gradients skipped (measurement build): the most a faster weight gradient can return
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 tools/probes/stream_tile_probe > gpurun_out/r04o_stream_probe.txt 2>&1
cat gpurun_out/r04o_stream_probe.txt
for skip in 0 1; do
  TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$PWD/transformertts_amd/lib/libttsmi_abl.so TTSMI_DEBUG_SKIP_WGRAD=$skip \
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip_wgrad', $skip, 'ms_per_step', d['ms_per_step'], 'host', d.get('host_issue_ms_per_step'))" \
    | tee -a gpurun_out/r04o_skip_wgrad.txt
done
