#!/bin/bash
# Round 6: mel filterbank as an in-register segmented scan - parity, then the mel leg against the previous library
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "mel or stft or audio or griffin" -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_mel_scan_tests.txt
OUT=$O/r06_mel_scan_ab.txt; : > $OUT
for i in 1 2 3; do
  for V in "" _melold; do
    TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=$R/transformertts_amd/lib/libttsmi$V.so timeout 300 python bench.py --workload mel --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('libttsmi$V mel GB/s', round(d['value'],1), 'ms', round(d['ms_per_step'],3))" | tee -a $OUT
  done
done
