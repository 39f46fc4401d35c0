#!/usr/bin/env python
"""Reference point only (not used by the product path): how fast do the vendor GEMMs (torch.matmul ->
hipBLASLt / rocBLAS) run the decoder shapes, bf16 in / bf16 out, fp32 accumulate, bias fused by addmm?
Usage (GPU box): python tools/probe_blas_reference.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_pitch import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    M = 28800
    bufs = 4
    for (K, N) in [(256, 1024), (256, 768), (1024, 256), (256, 256), (512, 256)]:
        w = (torch.randn(K, N, device=dev) * 0.05).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(bufs)]
        Os = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(bufs)]
        i = [0]

        def run():
            j = i[0] % bufs
            i[0] += 1
            torch.addmm(b, As[j], w, out=Os[j])
        t = timeit(run, n=40)
        byt = M * K * 2 + M * N * 2 + 2 * K * N
        print(f'vendor addmm K={K:5d} N={N:5d} bf16->bf16 {t:7.1f} us {byt / t / 1e6:6.2f} TB/s')


if __name__ == '__main__':
    main()
