#!/bin/bash
# the pre-pack's event is also waited for by an in-line repack: model-level chain test, ragged steps (plans whose shape falls under
# the threshold after a pre-pack), the step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
OUT=$O/r05_chain_prepack_check.txt; : > $OUT
timeout 120 python -m pytest tests/test_chain_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x -k "chained_blocks or train_step or ragged or bucket" 2>&1 | grep -E "passed|failed" | tee -a $OUT
TTSMI_DENSE_CHAIN_MIN_ROWS=12000 timeout 100 python bench.py --workload lj-dist --steps 60 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lj-dist (threshold 12000: chains on some shapes only) ms_per_step', round(d['ms_per_step'],3), 'loss', d['config'].get('loss_after'))" | tee -a $OUT
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-attention-maps --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('configs[1] ms_per_step', round(d['ms_per_step'],3), 'loss', d['config'].get('loss_after'))" | tee -a $OUT
