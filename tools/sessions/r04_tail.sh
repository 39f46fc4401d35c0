#!/bin/bash
# round 4: the part of the GPU suite that the aborted whole-suite run did not reach, on the final build, + smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 100 python -X faulthandler -m pytest tests/test_ops_gpu.py tests/test_reference_fixtures.py tests/test_reference_source_run.py tests/test_zz_reference_source_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror|Fatal" | tail -4 | tee gpurun_out/r04_tail_tests.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04_smoke.txt
