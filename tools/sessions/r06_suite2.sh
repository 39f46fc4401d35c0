#!/bin/bash
# the whole GPU suite twice on the final tree (after the widened wall-clock bound in test_griffinlim.py)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  ( time timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|^Extension modules\|amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 ) > $O/r06_full_gpu_tests_$i.txt 2>&1
  tail -4 $O/r06_full_gpu_tests_$i.txt
done
