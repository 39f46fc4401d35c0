#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['host_issue_ms_per_step'],3))"; }
for i in 1 2; do run; run --graph; done
