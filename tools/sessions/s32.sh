#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "attention or predict or val_step or maps or oracle" 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['ms_per_step_with_attention_maps'],3))"
python bench.py --workload predict 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for c in d['cases']: print(c['batch'], c['hipgraph'], c['attention_maps'], round(c['p50_ms'],3))"
