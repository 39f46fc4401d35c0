#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest split attention + model + config1"
timeout 900 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "split_key or bf16_attention" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py -q -m gpu 2>&1 | tail -5
echo "== predict probes (with maps is default in the probe)"
python bench.py --workload predict 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for c in d['cases']: print(c['batch'], c['hipgraph'], c['attention_maps'], round(c['p50_ms'],3), round(c['p90_ms'],3))"
echo "-- split off"
TTSMI_ATTN_SPLIT=0 python bench.py --workload predict 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for c in d['cases']: print(c['batch'], c['hipgraph'], c['attention_maps'], round(c['p50_ms'],3), round(c['p90_ms'],3))"
echo "-- fused LN at every M"
TTSMI_FUSE_LN_MIN_ROWS=0 python bench.py --workload predict 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
for c in d['cases']: print(c['batch'], c['hipgraph'], c['attention_maps'], round(c['p50_ms'],3), round(c['p90_ms'],3))"
