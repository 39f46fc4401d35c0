#!/usr/bin/env python
"""What does ONE kernel launch cost the host on this box, and how much of it is Python?  The ragged-batch workload
(bench.py --workload lj-dist) is bound by the host's issue time (3.7 ms per step for ~290 launches = 12.7 us each), so the
split decides what to attack: (a) the raw ctypes call of the smallest entry point (ttsmi_step_increment: one 1-thread
kernel), (b) the same through the ops wrapper (pointer / stream look-ups, error check), (c) a torch elementwise op,
(d) an autograd Function wrapped around (b).  Host time per call, queue kept shallow (sync every 200 calls)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from transformertts_amd import _lib, ops  # noqa: E402


def per_call(fn, n=2000, chunk=200):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(n // chunk):
        t0 = time.perf_counter()
        for _ in range(chunk):
            fn()
        t += time.perf_counter() - t0
        torch.cuda.synchronize()
    return 1e6 * t / n


def main():
    dev = torch.device('cuda:0')
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    l = _lib.lib()
    ptr, st = step.data_ptr(), torch.cuda.current_stream().cuda_stream
    raw = l.ttsmi_step_increment
    x = torch.zeros(64, device=dev)

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            ops.step_increment(step)
            return a

        @staticmethod
        def backward(ctx, g):
            ops.step_increment(step)
            return g

    a = torch.zeros(4, device=dev, requires_grad=True)
    ev = torch.cuda.Event()
    rows = [
        ('raw ctypes call (ttsmi_step_increment)', lambda: raw(ptr, st)),
        ('ops.step_increment (wrapper: _p, _stream, check)', lambda: ops.step_increment(step)),
        ('torch: x.add_(1)', lambda: x.add_(1)),
        ('torch.empty(1024) (caching allocator)', lambda: torch.empty(1024, device=dev)),
        ('torch.cuda.current_stream().cuda_stream', lambda: torch.cuda.current_stream().cuda_stream),
        ('event record + stream wait', lambda: (ev.record(), torch.cuda.current_stream().wait_event(ev))),
        ('autograd Function forward (1 launch)', lambda: Fn.apply(a)),
    ]
    for name, fn in rows:
        print(f'{name:55s} {per_call(fn):7.2f} us per call')
    y = Fn.apply(a).sum()
    t0 = time.perf_counter()
    for _ in range(500):
        y = Fn.apply(Fn.apply(Fn.apply(Fn.apply(a)))).sum()
        y.backward()
    torch.cuda.synchronize()
    print(f'{"4 Functions forward + sum + backward (10 launches)":55s} {1e6 * (time.perf_counter() - t0) / 500:7.2f} us per iteration')


if __name__ == '__main__':
    main()
