#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
run() { ( cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])" ); }
for i in 1 2 3; do run $R/_ab/old; run $R; done
