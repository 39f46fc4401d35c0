#!/usr/bin/env python
"""Host time of the C-issued train step: seconds per step inside the three C calls against the whole train_step call, for
the max-shape batch and for a sequence of changing shapes (what the bucketed producer brings), plus a cProfile of the latter."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from transformertts_amd import _lib  # noqa: E402
from transformertts_amd.model.models import ForwardTransformer  # noqa: E402
from transformertts_amd.utils.synthetic import synthetic_batch  # noqa: E402

cfg, shape = bench.workload_config('configs[1]')
model = ForwardTransformer.from_config(dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, precision='bf16', seed=0))
model._compile(learning_rate=1e-4)
dev = model.device
mk = lambda B, Tp, Tm, s: [torch.from_numpy(a).to(dev) for a in synthetic_batch(B, Tp, Tm, seed=s)]
big = mk(shape['B'], shape['Tp'], shape['Tm'], 1)
rng = np.random.default_rng(0)
ragged = [mk(int(b), int(tp), int(tm), 10 + i) for i, (b, tp, tm) in enumerate(
    (rng.integers(12, 40), rng.integers(100, 200), rng.integers(300, 890)) for _ in range(40))]
for b in [big] * 3 + ragged[:5]:
    model.train_step(*b)
torch.cuda.synchronize()
l = _lib.lib()
inner = [0.0]
orig = l._fns['ttsmi_ft_train_step']


def timed(*a):
    t0 = time.perf_counter()
    rc = orig(*a)
    inner[0] += time.perf_counter() - t0
    return rc


l._fns['ttsmi_ft_train_step'] = timed
for name, batches in (('max shape', [big] * 40), ('changing shapes', ragged)):
    inner[0] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        model.train_step(*b)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n = len(batches)
    print(f'{name:16s}: host {1e3 * host / n:.3f} ms/step, inside the C calls {1e3 * inner[0] / n:.3f}, wall {1e3 * wall / n:.3f}')
l._fns['ttsmi_ft_train_step'] = orig
pr = cProfile.Profile()
pr.enable()
for b in ragged:
    model.train_step(*b)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
