#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( for kv in TTSMI_DENSE_CHAIN=1 TTSMI_DENSE_CHAIN=0 TTSMI_DENSE_CHAIN=1 TTSMI_DENSE_CHAIN=0; do
  env $kv timeout 300 python bench.py --workload predict 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv', [(c['batch'], c['hipgraph'], c['attention_maps'], round(c['p50_ms'], 3)) for c in d['cases']])"
done ) > gpurun_out/r05t_predict_ab.txt 2>&1
cat gpurun_out/r05t_predict_ab.txt
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "predict" 2>&1 | tail -3
