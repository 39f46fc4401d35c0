#!/bin/bash
# where should the chains start?  the host-bound ragged workload and the headline with the row threshold at 0 / 8192 / 16384
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_chain_min_rows_ab.txt
: > $out
for thr in 16384 0 8192 16384 0; do
  env TTSMI_DENSE_CHAIN_MIN_ROWS=$thr timeout 300 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist min_rows $thr ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'real frames/s', round(d['value']))" | tee -a $out
done
for thr in 16384 0 4096; do
  env TTSMI_DENSE_CHAIN_MIN_ROWS=$thr timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('configs[1] min_rows $thr ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3))" | tee -a $out
done
