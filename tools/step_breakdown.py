#!/usr/bin/env python
"""Per-launch breakdown of one configs[1] train step: every C-ABI call bracketed by HIP events,
grouped by (entry point, shape).  With --no-overlap the wgrad side stream is disabled so each number
is the kernel alone on the GPU (its own roofline position, no co-running kernel).
Usage (GPU box): python tools/step_breakdown.py [--no-overlap] [--precision bf16|f32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-overlap', action='store_true')
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--min-us', type=float, default=15.0)
    args = ap.parse_args()
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch
    cfg, shape = bench.workload_config('configs[1]')
    cfg = dict(cfg, dropout_rate=0.1, predictors_dropout=0.1, device='cuda:0', seed=0, precision=args.precision,
               overlap_wgrad=not args.no_overlap)
    model = ForwardTransformer.from_config(cfg)
    model._compile(learning_rate=1e-4)
    batch = [torch.from_numpy(a).to('cuda:0') for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1)]
    for _ in range(3):
        model.train_step(*batch)
    torch.cuda.synchronize()
    agg = {}
    reps = 3
    for _ in range(reps):
        for fam, name, fl, by, key, ms in bench.instrumented_step(lambda: model.train_step(*batch)):
            a = agg.setdefault((name, key), [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += by; a[3] += ms
    tot = sum(a[3] for a in agg.values()) / reps
    print(f'sum of launch times {tot:.3f} ms/step')
    print(f'{"entry":28s} {"n/step":>6s} {"avg us":>8s} {"ms/step":>8s} {"MB":>7s} {"TB/s":>6s} {"TF/s":>7s}  shape')
    for (name, key), (n, fl, by, ms) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        if ms / n * 1e3 < args.min_us and ms / reps < 0.05:
            continue
        print(f'{name[6:]:28s} {n // reps:6d} {ms / n * 1e3:8.1f} {ms / reps:8.3f} {by / n / 1e6:7.1f} '
              f'{by / ms / 1e9:6.2f} {fl / ms / 1e9:7.1f}  {key}')


if __name__ == '__main__':
    main()
