#!/bin/bash
# round 4, session az: workgroup count of the mel launch at four resident workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
: > gpurun_out/r04az_mel_wgs.txt
for w in 4096 6144 8192 10240 12288 16384 24576; do
TTSMI_MEL_WGS=$w timeout 300 python bench.py --workload mel --no-cpu-baseline 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TTSMI_MEL_WGS $w GB/s', round(d['value'], 1), 'ms', round(d['ms_per_step'], 3))" | tee -a gpurun_out/r04az_mel_wgs.txt
done
