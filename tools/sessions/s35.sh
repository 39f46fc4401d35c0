#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "attention or keep_bit or conv or reference_default or full_depth or reproducible" 2>&1 | tail -4
python bench.py --workload ref-default --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ref-default', round(d['ms_per_step'],3))"
TTSMI_ATTN_DROPBITS=0 python bench.py --workload ref-default --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ref-default hashed', round(d['ms_per_step'],3))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('configs[1]', round(d['ms_per_step'],3))"
