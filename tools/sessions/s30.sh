#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), d['config']['loss_after'])"; }
for i in 1 2 3; do run A=1; run TTSMI_WGRAD_2STREAMS=1; run TTSMI_WGRAD_2STREAMS=1 TTSMI_WGRAD_WGS=64; done
