#!/usr/bin/env python
"""Host-side profile of the configs[1] train step (where do the ~3.4 ms of Python / ctypes per step go?):
cProfile over N steady-state steps, sorted by own time and by cumulative time."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from transformertts_amd.model.models import ForwardTransformer  # noqa: E402
from transformertts_amd.utils.synthetic import synthetic_batch  # noqa: E402

B = int(os.environ.get('B', '32'))
cfg, shape = bench.workload_config('configs[1]')
model = ForwardTransformer.from_config(dict(cfg, device='cuda:0', seed=0, precision='bf16'))
model._compile(learning_rate=1e-4)
batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(B, shape['Tp'], shape['Tm'], seed=1234)]
for _ in range(5):
    model.train_step(*batch)
torch.cuda.synchronize()
N = 40
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    model.train_step(*batch)
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    print(f'==== by {key} (totals over {N} steps)')
    pstats.Stats(pr).sort_stats(key).print_stats(38)
