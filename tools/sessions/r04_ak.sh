#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_griffinlim.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r04ak_tests.txt
cat gpurun_out/r04ak_tests.txt
