#!/bin/bash
# First session of the next round: (1) the whole GPU suite twice on the round-4 final build - the intermittent abort of
# round 4 (an out-of-bounds read of the attention forward / dQ kernels, fixed in its last hour) was only re-tested in parts;
# (2) same-box A/B of the two host-side knobs left opt-in / unmeasured at the end of round 4.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
  ( time timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|^Extension modules" | tail -15 ) > gpurun_out/r05_full_gpu_tests_$i.txt 2>&1
  tail -4 gpurun_out/r05_full_gpu_tests_$i.txt
done
: > gpurun_out/r05_host_knobs_ab.txt
for i in 1 2; do for kv in A=1 TTSMI_BWD_SAME_THREAD=1; do
  env $kv timeout 600 python bench.py --workload lj-dist 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lj-dist $kv ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3))" | tee -a gpurun_out/r05_host_knobs_ab.txt
done; done
