#!/usr/bin/env python
"""BASELINE.json configs[3]: mel feature-extraction microbench.  LJSpeech-length synthetic clips
(lengths ~ clip(N(145000, 48000^2), 24000, 222000), generated on the device), 22.05 kHz -> 1024-pt
STFT -> 80-bin log-mel in one batched launch.  Reports clips/s, frames/s and GB/s against the HBM
roofline with algorithmic bytes/clip = 4*N + 4*80*(1 + N//256).
--host adds the PCIe-inclusive rate: the same clips start in pinned HOST memory and the log-mel ends in
pinned host memory, chunked and pipelined over three streams (H2D copy, kernel, D2H copy).
Usage (GPU box): python tools/bench_mel.py [--clips 2000] [--cpu] [--host [--chunk 250]]"""
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
    ap.add_argument('--host', action='store_true', help='also time host wav -> host mel over PCIe (pipelined)')
    ap.add_argument('--chunk', type=int, default=250)
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
    if args.host:
        hw = wav.cpu().pin_memory()
        hm = torch.empty((frames, 80), dtype=torch.float32).pin_memory()
        groups = [(a, min(a + args.chunk, args.clips)) for a in range(0, args.clips, args.chunk)]
        tabs = []
        for a, b in groups:                       # per-chunk offset tables: a per-dataset constant, built untimed
            co = (clip_off[a:b + 1] - clip_off[a]).astype(np.int64)
            fo = (np.asarray(off[a:b + 1]) - int(off[a])).astype(np.int64)
            tabs.append((torch.from_numpy(co).cuda(), torch.from_numpy(fo).cuda(), int(fo[-1]),
                         torch.empty(int(co[-1]), dtype=torch.float32, device='cuda')))
        h2d, d2h, comp = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.current_stream()

        def pipeline():
            keep = []
            evs = []
            for (a, b), (co, fo, nf, dw) in zip(groups, tabs):
                with torch.cuda.stream(h2d):
                    dw.copy_(hw[int(clip_off[a]):int(clip_off[b])], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(h2d)
                    evs.append(ev)
            for (a, b), (co, fo, nf, dw), ev in zip(groups, tabs, evs):
                comp.wait_event(ev)
                m = ops.stft_logmel(dw, co, fo, nf, 1024, 256, audio._window, 80, lo, cnt, ptr, w, 0, 1e-5)
                done = torch.cuda.Event()
                done.record(comp)
                with torch.cuda.stream(d2h):
                    d2h.wait_event(done)
                    hm[int(off[a]):int(off[b])].copy_(m, non_blocking=True)
                keep.append(m)
            torch.cuda.synchronize()
            return keep

        pipeline()
        t0 = time.perf_counter()
        pipeline()
        dt = time.perf_counter() - t0
        assert os.environ.get("TTSMI_MEL_ABLATE") or torch.equal(hm, mel.cpu())
        out['host_to_host'] = {'ms': dt * 1e3, 'clips_per_s': args.clips / dt, 'chunk_clips': args.chunk,
                               'h2d_gb': 4.0 * total / 1e9, 'd2h_gb': 4.0 * 80 * frames / 1e9,
                               'pcie_gbs_in_plus_out': byt / dt / 1e9,
                               'note': 'pinned host wav -> H2D stream -> kernel -> D2H stream -> pinned host mel, '
                                       'wall clock of the whole pipeline; never the headline value'}
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
