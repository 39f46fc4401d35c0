#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== fused LN op tests"; timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_layernorm" 2>&1 | tail -12
echo "== model tests"; timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_config1_parity_gpu.py tests/test_dp_gloo.py -x -q -m gpu 2>&1 | grep -v "config1 parity" | tail -8
echo "== A/B"; python - <<'PY'
import torch, bench, time
from transformertts_amd.model.models import ForwardTransformer
from transformertts_amd.utils.synthetic import synthetic_batch
cfg, shape = bench.workload_config('configs[1]')
batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
variants = {'base(unplanned)': dict(planned_blocks=False), 'planned': dict(fuse_ln=False), 'planned+fuse_ln': dict()}
models = {}
for k, kw in variants.items():
    m = ForwardTransformer.from_config(dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16', **kw))
    m._compile(learning_rate=1e-4); models[k] = m
    for _ in range(4): out = m.train_step(*batch)
    print(k, 'loss', float(out['loss']))
torch.cuda.synchronize()
for rnd in range(3):
    for k, m in models.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.train_step(*batch)
        torch.cuda.synchronize(); print(k, round((time.perf_counter() - t0) / 20 * 1e3, 3), 'ms')
PY
