#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for d in $R/_ab/prev $R; do echo "== $d"; ( cd $d && python tools/kbench.py --only rowgemm 2>&1 | grep -v amdgpu.ids | grep "28800\|6400" | grep "fused\|xhat" ; python tools/kbench.py --only ln 2>&1 | grep -v amdgpu.ids | grep "ln  ") ; done
run() { ( cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))" ); }
for i in 1 2 3; do run $R/_ab/prev; run $R; done
