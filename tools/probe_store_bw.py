#!/usr/bin/env python
"""How fast can HBM be WRITTEN from a plain kernel?  (fill / copy of the FFN1 output size, rotating buffers)"""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (59, 118, 472):
    n = mb * 1000 * 1000 // 2
    xs = [torch.empty(n, dtype=torch.bfloat16, device='cuda') for _ in range(6)]
    ys = [torch.randn(n, device='cuda').bfloat16() for _ in range(2)]
    i = [0]
    def fill():
        xs[i[0] % 6].zero_(); i[0] += 1
    def copy():
        xs[i[0] % 6].copy_(ys[i[0] % 2]); i[0] += 1
    a, b = t(fill), t(copy)
    print(f'{mb} MB: fill {a:.1f} us = {mb / a * 1e-3 * 1e3:.2f} TB/s written; copy {b:.1f} us = {2 * mb / b:.2f} TB/s moved')
