#!/bin/bash
# weight-gradient launch width beside the chain kernels (TTSMI_WGRAD_WGS: target workgroups per launch; default 128)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
OUT=$O/r05_wgrad_wgs_ab.txt; : > $OUT
for w in 128 64 96 192 128; do
TTSMI_WGRAD_WGS=$w python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[1] wgrad target $w ms_per_step', round(d['ms_per_step'],3), 'host', round(d.get('host_issue_ms_per_step') or 0,3))" | tee -a $OUT
done
