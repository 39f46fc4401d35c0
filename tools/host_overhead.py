#!/usr/bin/env python
"""Host launch overhead of one train step: the configs[1] architecture at a tiny batch (the GPU work
is negligible, so the step time is the Python/ctypes/autograd launch loop).  Usage: python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch
    cfg, _ = bench.workload_config('configs[1]')
    for graph in (False, True):
        c = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision='bf16', use_graph=graph)
        m = ForwardTransformer.from_config(c)
        m._compile(learning_rate=1e-4)
        batch = [torch.from_numpy(a).cuda() for a in synthetic_batch(2, 16, 64, seed=1)]
        for _ in range(5):
            m.train_step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            m.train_step(*batch)
        torch.cuda.synchronize()
        print(f'use_graph={graph}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step at B=2 x 16 phonemes x 64 frames')


if __name__ == '__main__':
    main()
