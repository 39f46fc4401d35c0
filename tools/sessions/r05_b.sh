#!/bin/bash
# Which form of the guard allocator is free of false positives on this ROCm: stress + two test files under each knob set.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05b_guard_variants.txt
: > $out
( timeout 300 python tools/guard_selftest.py ) >> $out 2>&1
for kv in "A=1" "TTSMI_GUARD_SYNC_ALLOC=1" "TTSMI_GUARD_KEEP_VA=1" "TTSMI_GUARD_KEEP_VA=1 TTSMI_GUARD_SYNC_ALLOC=1" "TTSMI_GUARD_NO_RELEASE=1"; do
  echo "=== $kv" >> $out
  ( env $kv timeout 300 python tools/guard_stress.py 300 2>&1 | grep -v "^guard_alloc: CANARY" | tail -4 ) >> $out 2>&1
  ( env $kv TTSMI_GUARD_ALLOC=1 timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -n 1 --max-worker-restart 4 -k "add_layernorm or lenreg or masks_embedding" 2>&1 | grep -v "CANARY" | tail -6 ) >> $out 2>&1
done
cat $out
