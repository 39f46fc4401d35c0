#!/usr/bin/env python
"""Measured error of ttsmi_griffinlim against the oracle per iteration count (+ timing of the GPU loop)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import griffinlim_oracle as go, mel_oracle as mo
from transformertts_amd.data.audio import Audio
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_griffinlim import _speechlike, HOP, WIN, NFFT, SR
au = Audio(sampling_rate=SR, n_fft=NFFT, mel_channels=80, hop_length=HOP, win_length=WIN, f_min=0, f_max=8000, normalizer='MelGAN')
for T, n_iter in [(61, 0), (61, 1), (61, 4), (200, 32), (9, 3), (900, 32)]:
    y = _speechlike(HOP * (T - 1), 5)
    S = np.abs(mo.stft(y, NFFT, HOP, WIN))
    ang0 = go.random_phases(S.shape, 11)
    t0 = time.perf_counter(); want, aw = go.griffinlim(S, n_iter=n_iter, hop_length=HOP, win_length=WIN, angles=ang0, return_angles=True); tc = time.perf_counter() - t0
    got, ag = au.griffinlim(S, ang0, n_iter=n_iter, return_angles=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): au.griffinlim(S, ang0, n_iter=n_iter)
    tg = (time.perf_counter() - t0) / 5
    big = np.abs(S) > 1e-3 * S.max()
    print(f'T={T} n_iter={n_iter}: wav err {np.abs(got - want).max() / np.abs(want).max():.2e}  phase err (energetic bins) {np.abs(ag - aw)[big].max():.2e}  oracle {tc * 1e3:.1f} ms  gpu call {tg * 1e3:.2f} ms')
