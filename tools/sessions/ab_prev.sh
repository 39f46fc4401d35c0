#!/bin/bash
# A/B of the working tree against a copy of HEAD built under _ab/prev: tests first, then kbench + alternating bench runs.
#   rm -rf _ab && mkdir -p _ab/prev && git archive HEAD | tar -x -C _ab/prev && (cd _ab/prev && python __graft_entry__.py)
#   gpurun -- 'bash tools/sessions/ab_prev.sh "<pytest -k expression>" <kbench --only>'
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
python -m pytest tests/test_ops_gpu.py tests/test_config1_parity_gpu.py tests/test_model_gpu.py -q -m gpu -x -k "${1:-gemm or layernorm or config1 or parity}" 2>&1 | tail -3
if [ -n "$2" ]; then
  for d in $R/_ab/prev $R; do echo "== $d"; ( cd $d && python tools/kbench.py --only $2 2>&1 | grep -v amdgpu.ids | grep "28800\|6400" ); done
fi
run() { ( cd $1 && python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))" ); }
for i in 1 2 3; do run $R/_ab/prev; run $R; done
