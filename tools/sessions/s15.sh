#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
echo "== gemm tests"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "gemm or linear or split_key" 2>&1 | tail -5
echo "== kbench small gemm"
python tools/kbench.py --only gemm-small --variants TTSMI_HGEMM_DEEP=0 base TTSMI_HGEMM_DEEP=2 --json $O/s15_kbench.jsonl 2>&1 | grep -v amdgpu.ids
echo "== predict"
python tools/probe_predict.py 60 2>/dev/null
TTSMI_HGEMM_DEEP=0 python tools/probe_predict.py 60 2>/dev/null
echo "== train step A/B"
for v in 0 1; do TTSMI_HGEMM_DEEP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('DEEP=$v', d['ms_per_step'])"; done
for v in 0 1; do TTSMI_HGEMM_DEEP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('DEEP=$v', d['ms_per_step'])"; done
