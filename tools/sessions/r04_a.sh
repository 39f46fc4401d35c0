#!/bin/bash
# round 4, session A: the full -m gpu suite + smoke on the first build of the round, the default bench line with its new
# `also` legs (f32 step, mel, predict), and the lj-dist workload.  ONE gpurun call.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rm -f $O/config1_parity.jsonl
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/r04a_gpu_tests.txt; tail -4 $O/r04a_gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/r04a_bench_bf16.json 2> $O/r04a_bench_bf16.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04a_bench_bf16.json'))
print('ms_per_step', d['ms_per_step'], 'host', d['host_issue_ms_per_step'])
for k,v in d.get('also',{}).items():
    print(k, json.dumps(v)[:600])
PY
timeout 600 python bench.py --workload lj-dist > $O/r04a_bench_ljdist.json 2> $O/r04a_bench_ljdist.err; echo lj rc=$?
cut -c1-1500 $O/r04a_bench_ljdist.json; tail -3 $O/r04a_bench_ljdist.err
cat $O/bf16_vs_f32_curve.json
