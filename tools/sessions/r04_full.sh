#!/bin/bash
# round 4: the whole GPU suite + smoke on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 2400 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^  File \"/usr/local/lib/python3.10/dist-packages\|^Extension modules" | tail -60 ) > gpurun_out/r04_full_gpu_tests.txt 2>&1
cat gpurun_out/r04_full_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r04_smoke.txt
# the ragged workload once more (its host-bound step time moves with the box: 3.4 - 3.7 ms)
for i in 1 2; do timeout 600 python bench.py --workload lj-dist 2>/dev/null > gpurun_out/r04_bench_ljdist_run$i.json; python -c "import json; d = json.loads(open('gpurun_out/r04_bench_ljdist_run$i.json').read().strip().splitlines()[-1]); print('lj-dist run $i ms_per_step', round(d['ms_per_step'], 3), 'host', round(d.get('host_issue_ms_per_step', 0), 3), 'real frames/s', round(d['value']))"; done
