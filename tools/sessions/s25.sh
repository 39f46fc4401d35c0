#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 1 0; do
rm -f gpurun_out/config1_parity.jsonl
TTSMI_LN_CHAIN=$v python -m pytest tests/test_config1_parity_gpu.py -q -m gpu -k "bf16" 2>&1 | tail -1
python - <<PY
import json
for l in open('gpurun_out/config1_parity.jsonl'):
    d=json.loads(l)
    print('chain=$v', d['batch'], d['precision'], d['grad_worst'], d['grad_vec_worst'], round(d['mel'],5))
PY
done
