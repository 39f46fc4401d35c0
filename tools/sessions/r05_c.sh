#!/bin/bash
# The whole GPU suite under the guard allocator (tests/conftest.py, TTSMI_GUARD_ALLOC=1), one xdist worker so that a test
# that faults costs its worker, not the run; then the allocator's own stress as the false-positive check of the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TTSMI_GUARD_LOG=$PWD/gpurun_out/r05c_guard_log.txt
: > $TTSMI_GUARD_LOG
( timeout 200 python tools/guard_stress.py 200 2>&1 | tail -2 ) > gpurun_out/r05c_guard_suite.txt 2>&1
( time TTSMI_GUARD_ALLOC=1 timeout 2400 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider -n 1 --max-worker-restart 12 -rs 2>&1 \
    | grep -v "dist-packages\|^Extension modules\|amdgpu.ids" | tail -120 ) >> gpurun_out/r05c_guard_suite.txt 2>&1
tail -40 gpurun_out/r05c_guard_suite.txt
grep -c CANARY $TTSMI_GUARD_LOG; grep CANARY $TTSMI_GUARD_LOG | sed 's/at 0x[0-9a-f]*//' | sort | uniq -c | sort -rn | head -20
