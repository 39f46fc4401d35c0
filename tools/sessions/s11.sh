#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
echo "== attention tests (dh 192)"; timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -4
echo "== model tests"; timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -6
echo "== bench default (with roofline)"; timeout 600 python bench.py > $O/s11_bench.json 2> $O/s11_bench.err; echo rc=$?; python - <<'PY'
import json
d=json.load(open('gpurun_out/s11_bench.json'))
print(d['ms_per_step'], d['value'], d.get('ms_per_step_with_attention_maps'), d['host_issue_ms_per_step'])
r=d['roofline']; print({k:r[k] for k in ('bound','kernel','achieved','peak','frac','avg_launch_ms','launches_per_step','c_abi_calls_per_step')})
for k,v in r['per_kernel'].items(): print(f"{v['ms']:.3f} ms {v['launches']:4d}  {k[:90]}")
PY
tail -3 $O/s11_bench.err
echo "== bench predict"; timeout 600 python bench.py --workload predict > $O/s11_predict.json 2> $O/s11_predict.err; echo rc=$?; python - <<'PY'
import json
d=json.load(open('gpurun_out/s11_predict.json'))
for c in d['cases']: print(c)
PY
tail -3 $O/s11_predict.err
echo "== bench ref-default"; timeout 600 python bench.py --workload ref-default --steps 5 --warmup 2 --no-cpu-baseline --no-attention-maps > $O/s11_refdefault.json 2> $O/s11_refdefault.err; echo rc=$?; cut -c1-600 $O/s11_refdefault.json; tail -3 $O/s11_refdefault.err
