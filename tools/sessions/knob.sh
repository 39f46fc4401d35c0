#!/bin/bash
# alternate bench runs of the default build and of environment-knob variants: bash tools/sessions/knob.sh VAR=1 "A=1 B=2" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run A=1; for v in "$@"; do run "$v"; done; done
