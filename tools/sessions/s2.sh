#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
echo "== ops tests (gemm)"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== kbench gemm"; timeout 900 python tools/kbench.py --only gemm --variants base TTSMI_HGEMM_DMA=0 TTSMI_HGEMM_DMA=1 > $O/s2_kbench_gemm.txt 2>&1; cat $O/s2_kbench_gemm.txt
echo "== bench A/B DMA"; for v in 3 0 1 3 0; do TTSMI_HGEMM_DMA=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DMA=$v', d['ms_per_step'])"; done
echo "== determinism"; timeout 300 python tools/check_determinism.py --once --steps 40 2>&1 | tail -3
