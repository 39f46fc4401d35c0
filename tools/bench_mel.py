#!/usr/bin/env python
"""BASELINE.json configs[3]: mel feature-extraction microbench.  LJSpeech-length synthetic clips
(lengths ~ clip(N(145000, 48000^2), 24000, 222000), generated on the device), 22.05 kHz -> 1024-pt
STFT -> 80-bin log-mel in one batched launch.  Reports clips/s, frames/s and GB/s against the HBM
roofline with algorithmic bytes/clip = 4*N + 4*80*(1 + N//256).
Usage (GPU box): python tools/bench_mel.py [--clips 2000] [--cpu]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=2000)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--cpu', action='store_true', help='also time the numpy oracle on a few clips')
    args = ap.parse_args()
    from transformertts_amd.data.audio import Audio
    audio = Audio(22050, 1024, 80, 256, 1024, 0, 8000, 'MelGAN')
    rng = np.random.default_rng(0)
    lens = np.clip(rng.normal(145000, 48000, args.clips), 24000, 222000).astype(np.int64)
    total = int(lens.sum())
    gen = torch.Generator(device='cuda').manual_seed(0)
    wav = torch.randn(total, device='cuda', generator=gen) * 0.1
    t = torch.arange(total, device='cuda', dtype=torch.float32) / 22050.0
    wav += 0.3 * torch.sin(2 * np.pi * 220.0 * t) + 0.2 * torch.sin(2 * np.pi * 1760.0 * t)
    del t
    mel, off = audio.mel_spectrogram_batch(wav, lens.tolist())
    torch.cuda.synchronize()
    frames = int(off[-1])
    # time the kernel with the offset tables already on the device (the tables are a per-dataset
    # constant; building them on the host is not part of the hot path)
    from transformertts_amd import ops
    clip_off = np.zeros(args.clips + 1, dtype=np.int64)
    clip_off[1:] = np.cumsum(lens)
    coff = torch.from_numpy(clip_off).cuda()
    foff = torch.from_numpy(np.asarray(off)).cuda()
    lo, cnt, ptr, w = audio._mel

    def run():
        return ops.stft_logmel(wav, coff, foff, frames, 1024, 256, audio._window, 80, lo, cnt, ptr, w, 0, 1e-5)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        mel2 = run()
    e1.record()
    torch.cuda.synchronize()
    assert os.environ.get("TTSMI_MEL_ABLATE") or torch.equal(mel2, mel)
    ms = e0.elapsed_time(e1) / args.reps
    byt = 4.0 * total + 4.0 * 80 * frames
    out = {'metric': 'mel extraction', 'clips': args.clips, 'frames': frames, 'ms': ms,
           'clips_per_s': args.clips / ms * 1e3, 'frames_per_s': frames / ms * 1e3,
           'roofline': {'bound': 'hbm', 'achieved': byt / ms / 1e6, 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': byt / ms / 1e6 / 8000.0, 'algorithmic_bytes': byt},
           'note': 'kernel time only (HIP events around the launches; offset tables resident)'}
    if args.cpu:
        from oracle import mel_oracle as mo
        n = 8
        clips = [wav[int(lens[:i].sum()):int(lens[:i + 1].sum())].cpu().numpy() for i in range(n)]
        t0 = time.perf_counter()
        for c in clips:
            mo.mel_spectrogram(c)
        dt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': n / dt, 'unit': 'clips/s', 'cores': 1, 'kind': 'port',
                               'sample': f'{n} clips, numpy rfft + dense mel restatement of data/audio.py:81-92'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
