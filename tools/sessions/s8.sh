#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== model tests"; timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py tests/test_dp_gloo.py -x -q -m gpu 2>&1 | grep -v "config1 parity" | tail -8
echo "== bench"; timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], 'host', d['host_issue_ms_per_step'])"
echo "== hostbound probe"; timeout 300 python tools/probe_hostbound.py 2>&1 | tail -4
