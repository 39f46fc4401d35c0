#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "attention" 2>&1 | tail -3
python tools/kbench.py --only attn 2>&1 | grep -v amdgpu.ids
run() { ( cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))" ); }
for i in 1 2 3; do run $R; run $R/_ab/occ; done
