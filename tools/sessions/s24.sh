#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -x 2>&1 | tail -4
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run A=1; run TTSMI_LN_CHAIN=0; done
