#!/usr/bin/env python
"""Host-side profile of the configs[1] train step (where do the ~3.4 ms of Python / ctypes per step go?):
cProfile over N steady-state steps, sorted by own time and by cumulative time.  SINGLE=1: the backward pass runs on the
calling thread (torch.autograd.set_multithreading_enabled(False)) so that the profiler - which only sees the thread it was
enabled on - also sees the Functions' backward bodies; the step is timed first with and without it."""
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
import contextlib  # noqa: E402
import time  # noqa: E402

for _ in range(5):
    model.train_step(*batch)
torch.cuda.synchronize()
N = 40
SINGLE = os.environ.get('SINGLE', '0') == '1'
for single in (False, True, False, True):
    ctxm = torch.autograd.set_multithreading_enabled(False) if single else contextlib.nullcontext()
    with ctxm:
        for _ in range(3):
            model.train_step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            model.train_step(*batch)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    print(f'backward on the {"calling" if single else "engine"} thread: host issue {1e3 * host / N:.3f} ms, wall {1e3 * wall / N:.3f} ms per step')
pr = cProfile.Profile()
with (torch.autograd.set_multithreading_enabled(False) if SINGLE else contextlib.nullcontext()):
    pr.enable()
    for _ in range(N):
        model.train_step(*batch)
    pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    print(f'==== by {key} (totals over {N} steps)')
    pstats.Stats(pr).sort_stats(key).print_stats(60)
