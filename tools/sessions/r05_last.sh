#!/bin/bash
# the round's last tree (the pre-pack is host code only: same library digest): the whole suite once more, then the default line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
( time timeout 900 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "dist-packages\|^Extension modules\|amdgpu.ids" | tail -12 ) > $O/r05_full_gpu_tests_3.txt 2>&1
tail -5 $O/r05_full_gpu_tests_3.txt
python bench.py > $O/r05_bench_bf16_final_tree.json 2>/dev/null; echo rc=$?
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05_bench_bf16_final_tree.json').read().strip().splitlines()[-1])
print(round(d['value'], 1), d['unit'], 'ms', round(d['ms_per_step'], 3), 'maps', d.get('ms_per_step_with_attention_maps'), 'traffic', (d.get('roofline') or {}).get('traffic'))
for k, v in (d.get('also') or {}).items():
    print('   also', k, {kk: vv for kk, vv in v.items() if kk in ('ms_per_step', 'value', 'error', 'host_issue_ms_per_step')})
PY
