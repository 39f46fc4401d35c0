#!/bin/bash
# round 4, session M: dK/dV no-padding fast path (+ the mel and rowgemm changes since HEAD~) - attention tests, kbench attn, step A/B vs _ab/prev
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu -k "attention or attn or keep_bit or one_pass" 2>&1 | tail -3
for d in $R/_ab/prev $R; do echo "== $d"; ( cd $d && python tools/kbench.py --only attn 2>&1 | grep -v amdgpu.ids | grep " 900    64\| 200    64" | grep "bwd" ); done
run() { ( cd $1 && python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['loss_after'])" ); }
for i in 1 2 3; do run $R/_ab/prev; run $R; done
