#!/usr/bin/env python
"""Does a power-of-two row pitch cost bandwidth (L2/HBM channel aliasing)?  Times the wgrad and forward
GEMM kernels on the decoder shapes with contiguous operands vs the same operands at a padded pitch
(+128 bytes per row).  Usage (GPU box): python tools/probe_pitch.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def padded(M, C, dtype, pad_bytes, dev):
    esz = 2 if dtype == torch.bfloat16 else 4
    pad = pad_bytes // esz
    buf = torch.randn(M, C + pad, device=dev).to(dtype)
    return buf[:, :C]


def main():
    dev = 'cuda:0'
    M = 28800
    for pad in (0, 128, 256, 64):
        print(f'--- row pitch padding {pad} B')
        for (K, N, xdt, ydt) in [(256, 1024, torch.float32, torch.bfloat16), (1024, 256, torch.bfloat16, torch.float32),
                                 (256, 256, torch.float32, torch.float32), (256, 768, torch.float32, torch.bfloat16)]:
            x = padded(M, K, xdt, pad, dev)
            dy = padded(M, N, ydt, pad, dev)
            dw = torch.empty(K, N, device=dev)
            db = torch.empty(N, device=dev)
            t = timeit(lambda: ops.hgemm_wgrad_rows(x, dy, dw, db))
            print(f'wgrad_rows K={K:5d} N={N:5d} x={str(xdt)[6:]:8s} dy={str(ydt)[6:]:8s} {t:7.1f} us')
        for (K, N, adt, obf, relu) in [(256, 1024, torch.float32, True, True), (256, 768, torch.float32, True, False),
                                       (1024, 256, torch.bfloat16, False, False), (256, 256, torch.float32, False, False)]:
            a = padded(M, K, adt, pad, dev)
            w = torch.randn(K, N, device=dev) * 0.05
            sh = ops.make_shadow(w)
            b = torch.randn(N, device=dev)
            out = padded(M, N, torch.bfloat16 if obf else torch.float32, pad, dev)
            t = timeit(lambda: ops.hgemm_tn(a, sh.wt, b, relu=relu, out=out))
            print(f'hgemm_tn   K={K:5d} N={N:5d} a={str(adt)[6:]:8s} out_bf16={obf!s:5s} {t:7.1f} us')


if __name__ == '__main__':
    main()
