#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|^Extension modules\|amdgpu.ids" | tail -25 ) > gpurun_out/r05r_full_gpu_tests.txt 2>&1
tail -8 gpurun_out/r05r_full_gpu_tests.txt
python bench.py > gpurun_out/r05r_bench.json 2>/dev/null
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05r_bench.json').read().strip().splitlines()[-1])
print('ms', d['ms_per_step'], 'maps', d['ms_per_step_with_attention_maps'], 'host', d['host_issue_ms_per_step'], 'kernel', d['roofline']['kernel'][:40], d['roofline']['frac'])
print({k: (v.get('ms_per_step'), v.get('error')) for k, v in d['also'].items()})
for k, v in d['roofline']['per_kernel'].items():
    print('  ', k[:60], v['launches'], round(v['ms'], 3))
PY
