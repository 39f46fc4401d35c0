#!/usr/bin/env python
"""The reference-default conv stacks' large GEMMs alone (plain-GEMM route of ops.ConvStackFn: overlapping A rows), whatever
kernel the library's router picks under the environment's knobs: us per launch and TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformertts_amd import _lib, ops  # noqa: E402

dev = 'cuda:0'
rows = 32 * 902
for name, C, N in (('conv1 fwd  (K 1152 -> N 1536)', 384, 1536), ('conv2 fwd  (K 4608 -> N 384)', 1536, 384),
                   ('conv2 dgrad (K 1152 -> N 1536)', 384, 1536), ('conv1 dgrad (K 4608 -> N 384)', 1536, 384)):
    K, M = 3 * C, rows - 2
    bufs = [torch.randn(rows + 2, C, device=dev).to(torch.bfloat16) for _ in range(3)]
    w = torch.randn(K, N, device=dev) * 0.03
    sh = ops.make_shadow(w)
    b = torch.randn(N, device=dev)
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    views = [torch.as_strided(x, (M, K), (C, 1)) for x in bufs]

    def run(i):
        ops.hgemm_tn(views[i % 3], sh.wt, b, relu=True, out=outs[i % 3])
    for i in range(4):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for i in range(n):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    kern = _lib.lib().ttsmi_last_kernel().decode()
    print(f'{name:34s} {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TF  {kern}')
