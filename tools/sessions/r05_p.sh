#!/bin/bash
# A/B of the backward pass on the calling thread (TTSMI_BWD_SAME_THREAD=1) on the host-bound ragged workload and the headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_bwd_same_thread_ab.txt
: > $out
for i in 1 2 3; do for kv in A=1 TTSMI_BWD_SAME_THREAD=1; do
  env $kv timeout 300 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist $kv ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3))" | tee -a $out
done; done
for i in 1 2; do for kv in A=1 TTSMI_BWD_SAME_THREAD=1; do
  env $kv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] $kv ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3))" | tee -a $out
done; done
