#!/bin/bash
# round 4, session K: 64 x 64 wave tiles in the 128-row full-row GEMM + LayerNorm kernels - parity, kernel timings, step A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "layernorm or rowgemm or fused_gemm or hgemm_ln or residual" 2>&1 | tail -4
for v in 1 0; do echo "== TTSMI_ROWGEMM_W64=$v"; TTSMI_ROWGEMM_W64=$v timeout 200 python tools/kbench.py --only rowgemm 2>&1 | grep -v amdgpu.ids | grep "28800" ; done
run() { env $1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run TTSMI_ROWGEMM_W64=1; run TTSMI_ROWGEMM_W64=0; done
