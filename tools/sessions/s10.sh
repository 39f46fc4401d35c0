#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== k256 tests (forced)"; TTSMI_HGEMM_K256=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k256 or hgemm" 2>&1 | tail -5
echo "== k256 tests (default route)"; timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "k256" 2>&1 | tail -3
echo "== kbench"; timeout 600 python tools/kbench.py --only gemm --variants TTSMI_HGEMM_K256=0 base 2>&1 | grep -E "kernel|qkv|ffn1|dctx"
echo "== model tests"; timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py -x -q -m gpu 2>&1 | grep -v "config1 parity" | tail -5
echo "== bench A/B"; for v in 0 2 0 2; do TTSMI_HGEMM_K256=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K256=$v', d['ms_per_step'])"; done
