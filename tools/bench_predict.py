#!/usr/bin/env python
"""BASELINE.json configs[4]: inference-only ForwardTransformer.predict, batch 1 and batch 64 long
sentences (400 phonemes), 1 GPU.  Durations are supplied (synthetic, mean 4.5 frames per phoneme, as
in LJSpeech) so that the decoder length is realistic with random-init weights; attention maps are not
materialised (model.return_attention = False: at 1800 frames they are 3.3 GB per layer at batch 64).
Reports p50 / p90 latency and RTF = latency / seconds of audio produced (hop 256 @ 22.05 kHz).
Eager launches (one host sync for the data-dependent mel length, like the reference's eager call): the
hipGraph replay measured on the train step was slower per node on ROCm 7.2 than the eager loop.
Usage (GPU box): python tools/bench_predict.py [--precision bf16|f32]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--phonemes', type=int, default=400)
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--cpu', action='store_true', help='also time the torch-CPU restatement (oracle) at batch 1')
    args = ap.parse_args()
    from transformertts_amd.model.models import ForwardTransformer
    cfg, _ = bench.workload_config('configs[1]')
    cfg = dict(cfg, device='cuda:0', seed=0, precision=args.precision)
    model = ForwardTransformer.from_config(cfg)
    model.return_attention = False
    rng = np.random.default_rng(1234)
    out = {'metric': 'predict latency', 'precision': args.precision, 'phonemes': args.phonemes, 'cases': []}
    for B in (1, 64):
        tok = rng.integers(1, 127, size=(B, args.phonemes)).astype(np.int32)
        dur = rng.multinomial(int(4.5 * args.phonemes), np.ones(args.phonemes) / args.phonemes, size=B).astype(np.int32)
        tok_d, dur_d = torch.from_numpy(tok).cuda(), torch.from_numpy(dur).cuda()
        fn = lambda: model.predict(tok_d, encode=False, phoneme_durations=dur_d)   # noqa: E731
        for _ in range(5):
            o = fn()
        torch.cuda.synchronize()
        lat = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            o = fn()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        frames = int(o['expanded_lengths'].sum().item())
        audio_s = frames * 256 / 22050.0
        lat = np.sort(np.array(lat))
        p50, p90 = float(lat[len(lat) // 2]), float(lat[int(len(lat) * 0.9)])
        out['cases'].append({'batch': B, 'frames': frames, 'audio_seconds': audio_s, 'p50_ms': p50 * 1e3,
                             'p90_ms': p90 * 1e3, 'rtf_p50': p50 / audio_s,
                             'mel_frames_per_s': frames / p50})
    if args.cpu:
        # reference-restatement CPU baseline (TF2 unavailable offline): oracle.call(training=False), fp32,
        # attention maps materialised like the reference, batch 1, same sentence shape
        from oracle import ft_oracle as fo
        torch.set_num_threads(bench.usable_cpus())
        ocfg = {k: v for k, v in cfg.items() if k not in ('device', 'seed', 'precision')}
        om = fo.ForwardTransformerOracle(ocfg, fo.init_weights(ocfg, seed=0), torch.float32)
        tok1 = rng.integers(1, 127, size=(1, args.phonemes)).astype(np.int32)
        dur1 = rng.multinomial(int(4.5 * args.phonemes), np.ones(args.phonemes) / args.phonemes, size=1).astype(np.int32)
        with torch.no_grad():
            om.call(tok1, target_durations=dur1[..., None])
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                o = om.call(tok1, target_durations=dur1[..., None])
            dt = (time.perf_counter() - t0) / n
        frames = int(4.5 * args.phonemes)
        out['cpu_baseline'] = {'batch': 1, 'latency_ms': dt * 1e3, 'rtf': dt / (frames * 256 / 22050.0),
                               'cores': bench.usable_cpus(), 'kind': 'port',
                               'sample': f'{n} calls of the torch-CPU fp32 restatement of predict (models.py:559-577)'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
