#!/bin/bash
# round 4, session n: the GPU NNLS (ttsmi_mel_nnls) against the oracle's scipy solution + reconstruct_waveform timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_griffinlim.py -x -q -m gpu -s 2>&1 | tail -25 > gpurun_out/r04n_tests.txt
cat gpurun_out/r04n_tests.txt
