#!/usr/bin/env python
"""Micro-benchmark of the GEMM entry points at the config-2 shapes (HIP events on the launch stream).
Usage (GPU box): python tools/bench_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import ops  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3     # us


def main():
    dev = 'cuda:0'
    shapes = [(28800, 256, 768), (28800, 512, 256), (28800, 256, 1024), (28800, 1024, 256), (6400, 256, 1024),
              (28800, 256, 80)]
    print(f'{"shape":>22s} {"f32 fwd":>9s} {"bf16 fwd":>9s} {"TF":>6s} {"GB/s":>6s} | {"bf16 dgrad":>10s} '
          f'{"bf16 wgrad":>10s} {"castT x":>8s} {"castT dy":>8s} {"wgrad_rows":>9s}')
    for M, K, N in shapes:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        dy = torch.randn(M, N, device=dev)
        sh = ops.make_shadow(w)
        t_f32 = timeit(lambda: ops.linear_fwd(x, w, b))
        t_h = timeit(lambda: ops.hgemm_tn(x, sh.wt, b))
        t_dg = timeit(lambda: ops.hgemm_tn(dy, sh.wb)) if N % 8 == 0 else float('nan')
        xT, dyT = ops.cast_transpose_bf16(x), ops.cast_transpose_bf16(dy)
        dw, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
        t_wg = timeit(lambda: ops.hgemm_wgrad(xT, dyT, dw, db, M))
        t_wr = timeit(lambda: ops.hgemm_wgrad_rows(x, dy, dw, db)) if N % 4 == 0 else float('nan')
        t_cx = timeit(lambda: ops.cast_transpose_bf16(x))
        t_cy = timeit(lambda: ops.cast_transpose_bf16(dy))
        fl = 2.0 * M * K * N
        byt = 4.0 * (M * K + M * N) + 2.0 * K * N
        print(f'{str((M, K, N)):>22s} {t_f32:9.1f} {t_h:9.1f} {fl / t_h / 1e6:6.0f} {byt / t_h / 1e3:6.0f} | '
              f'{t_dg:10.1f} {t_wg:10.1f} {t_cx:8.1f} {t_cy:8.1f} {t_wr:9.1f}')


if __name__ == '__main__':
    main()
