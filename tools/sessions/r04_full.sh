#!/bin/bash
# round 4: the whole GPU suite + smoke on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 2400 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^  File \"/usr/local/lib/python3.10/dist-packages\|^Extension modules" | tail -60 ) > gpurun_out/r04_full_gpu_tests.txt 2>&1
cat gpurun_out/r04_full_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r04_smoke.txt
