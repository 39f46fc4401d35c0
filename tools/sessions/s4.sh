#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 32 64 16; do timeout 300 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=$b', d['ms_per_step'], 'host', d['host_issue_ms_per_step'])"; done
timeout 300 python bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'], 'host', d['host_issue_ms_per_step'])"
