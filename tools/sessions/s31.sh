#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3))"; }
for i in 1 2; do run A=1; run TTSMI_DEBUG_SKIP_WGRAD=1; run TTSMI_WGRAD_WGS=256; run TTSMI_WGRAD_WGS=192; done
python tools/probe_phases.py 2>&1 | grep -v amdgpu.ids | head -7
