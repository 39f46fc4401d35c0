#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== model tests"; timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py tests/test_dp_gloo.py tests/test_zz_reference_source_gpu.py -x -q -m gpu 2>&1 | tail -8
python - <<'PY'
import torch, bench, time
from transformertts_amd.model.models import ForwardTransformer
from transformertts_amd.utils.synthetic import synthetic_batch
cfg, shape = bench.workload_config('configs[1]')
batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
models = {}
for ov in (False, True):
    m = ForwardTransformer.from_config(dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16', overlap_predictors=ov))
    m._compile(learning_rate=1e-4); models[ov] = m
    for _ in range(4): m.train_step(*batch)
torch.cuda.synchronize()
for rnd in range(3):
    for ov in (False, True):
        m = models[ov]; torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.train_step(*batch)
        torch.cuda.synchronize(); print('overlap_predictors', ov, round((time.perf_counter() - t0) / 20 * 1e3, 3), 'ms')
print('params equal:', torch.equal(models[False].params.data, models[True].params.data))
PY
