#!/usr/bin/env python
"""Forward-GEMM variants at the decoder shapes: fp32 vs bf16 A operand, bf16 vs fp32 output.
Usage (GPU box): [TTSMI_HGEMM_BM=64|128] python tools/probe_gemm_variants.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import ops  # noqa: E402
from tools.probe_pitch import timeit  # noqa: E402


def main():
    dev = 'cuda:0'
    M = 28800
    bufs = 4                                   # rotate buffers so the working set exceeds the Infinity Cache
    for (K, N) in [(256, 1024), (256, 768), (1024, 256), (256, 256), (512, 256)]:
        w = torch.randn(K, N, device=dev) * 0.05
        sh = ops.make_shadow(w)
        b = torch.randn(N, device=dev)
        for adt in (torch.float32, torch.bfloat16):
            for obf in (True, False):
                As = [torch.randn(M, K, device=dev).to(adt) for _ in range(bufs)]
                Os = [torch.empty(M, N, device=dev, dtype=torch.bfloat16 if obf else torch.float32) for _ in range(bufs)]
                i = [0]

                def run():
                    j = i[0] % bufs
                    i[0] += 1
                    ops.hgemm_tn(As[j], sh.wt, b, out=Os[j])
                t = timeit(run, n=40)
                byt = M * K * As[0].element_size() + M * N * Os[0].element_size() + 2 * K * N
                print(f'K={K:5d} N={N:5d} A={str(adt)[6:]:8s} out={"bf16" if obf else "f32 "} {t:7.1f} us '
                      f'{byt / t / 1e6:6.2f} TB/s')


if __name__ == '__main__':
    main()
