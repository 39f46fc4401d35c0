#!/bin/bash
# Round 6: the 64-row (four-wave) forms of the chain kernels - parity, then alone against the launches they replace and
# against the 128-row form, at encoder / bucketed-batch row counts
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_chain_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | head -10 | tee $O/r06_chain64_tests.txt
OUT=$O/r06_chain64_alone.txt; : > $OUT
for nw in 0 8; do
  echo "== TTSMI_DENSE_CHAIN_NW=$nw (0 = by row count: 4 waves up to 16 384 rows)" | tee -a $OUT
  TTSMI_DENSE_CHAIN_NW=$nw python tools/bench_chain.py 28800 12000 9000 6400 4000 2000 2>&1 | grep "M=" | tee -a $OUT
  TTSMI_DENSE_CHAIN_NW=$nw python tools/bench_chain_bwd.py 28800 12000 9000 6400 4000 2000 2>&1 | grep -i "M=" | tee -a $OUT
done
