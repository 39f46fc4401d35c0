#!/bin/bash
# Round 5, first session: (1) the guard allocator's self test (a read past a tensor's end faults; caching allocator: unnoticed),
# (2) the whole GPU suite under the guard allocator (tests/conftest.py, TTSMI_GUARD_ALLOC=1; xdist so that a faulting test
# kills one worker, not the run), (3) the default bench line (maps-returning step on ring buffers; lj_dist / ref_default legs).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TTSMI_GUARD_LOG=$PWD/gpurun_out/r05a_guard_log.txt
: > $TTSMI_GUARD_LOG
( timeout 300 python tools/guard_selftest.py ) > gpurun_out/r05a_guard_selftest.txt 2>&1
cat gpurun_out/r05a_guard_selftest.txt
( time TTSMI_GUARD_ALLOC=1 timeout 1500 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider -n 2 --max-worker-restart 12 -rs 2>&1 \
    | grep -v "dist-packages\|^Extension modules" | tail -60 ) > gpurun_out/r05a_guard_suite.txt 2>&1
tail -25 gpurun_out/r05a_guard_suite.txt
sort $TTSMI_GUARD_LOG | uniq -c | sort -rn | head -20
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05a_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'maps', d['ms_per_step_with_attention_maps'], d.get('attention_maps_allocator'), 'host', d['host_issue_ms_per_step'])
for k, v in d.get('also', {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ('ms_per_step', 'value', 'error', 'leg_seconds', 'host_issue_ms_per_step')})
PY
tail -3 gpurun_out/r05a_bench.err
