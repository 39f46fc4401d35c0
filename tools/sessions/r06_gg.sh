#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
L=$R/transformertts_amd/lib
for V in "" _x3o4 _x3o2 "" _x3o4; do TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$L/libttsmi$V.so timeout 300 python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('libttsmi$V bf16x3 ms_per_step', round(d['ms_per_step'],3))" | tee -a $O/r06_x3_occupancy.txt; done
