#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_cstep_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
python tools/debug/cstep_host.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_cstep_host.txt | head -70
python bench.py --workload lj-dist --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('lj-dist ms_per_step', d['ms_per_step'], 'host', d['host_issue_ms_per_step'], 'stall', d['host_stall_ms_per_step'], 'ratio', d['ragged_over_max_shape_per_padded_frame'])" | tee -a $O/r06_cstep_host.txt
