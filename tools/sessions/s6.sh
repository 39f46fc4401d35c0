#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== attention tests"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -8
echo "== model tests"; timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py -x -q -m gpu 2>&1 | tail -6
echo "== bench A/B dropbits"; for v in 1 0 1 0; do TTSMI_ATTN_DROPBITS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DROPBITS=$v', d['ms_per_step'])"; done
