"""CPU restatement ORACLE of Audio.reconstruct_waveform (data/audio.py:94-110): normalised mel -> linear magnitudes
(librosa.feature.inverse.mel_to_stft) -> Griffin-Lim (librosa.core.griffinlim).

TEST INFRASTRUCTURE ONLY (see oracle/ft_oracle.py header): imported by tests/ only, never by the product.
PARITY UNPINNED: the arithmetic lives in librosa==0.7.1 (requirements.txt:2), which is neither installed nor vendored
under /root/reference, and the reference's own call draws its start phases from the UNSEEDED global NumPy generator
(random_state=None), so even the reference cannot reproduce its own output.  The published 0.7.1 algorithm is restated
below [3P], function by function, with the dtypes NumPy's promotion rules give the original expressions:

  librosa.feature.inverse.mel_to_stft(M, sr, n_fft, power=1, fmin, fmax)      (call site data/audio.py:98-104)
      mel_basis = filters.mel(sr, n_fft, n_mels=M.shape[0], dtype=M.dtype, fmin, fmax)
      inverse   = util.nnls(mel_basis, M);   inverse ** (1 / power)
  librosa.util.nnls(A, B)  (B 2-D)  blocks of MAX_MEM_BLOCK // (A.shape[-1] * A.itemsize) columns (127 for a fp32 513-bin
      basis); start x = clip(lstsq(A, B), 0); per block scipy.optimize.fmin_l_bfgs_b on
      f(x) = 0.5 ||A x - B||^2,  grad = A^T (A x - B),  bounds x >= 0,  m = A.shape[1] corrections
  librosa.core.griffinlim(S, n_iter=32, hop_length, win_length)               (call site data/audio.py:105-109)
      momentum 0.99, init 'random': angles (complex64) = exp(2j pi rng.rand(*S.shape));  rebuilt = 0
      loop: inverse = istft(S * angles); tprev = rebuilt; rebuilt = stft(inverse);
            angles = rebuilt - momentum / (1 + momentum) * tprev;  angles /= |angles| + 1e-16
      return istft(S * angles)
  librosa.core.istft: irfft (fp64) * window, overlap-add into a fp32 signal in frame order, division by
      filters.window_sumsquare where it exceeds util.tiny (fp32), n_fft // 2 trimmed from both ends (center=True)

Anchors without librosa (tests/test_griffinlim.py): stft(istft(X)) round trips, scipy.signal.istft on the same frames,
the NOLA envelope of the periodic Hann at hop = n_fft / 4 (exactly 1.5 inside the signal), the NNLS optimality
conditions (KKT) of the returned solution, and spectral convergence on a signal whose spectrogram is consistent."""
from __future__ import annotations

import numpy as np
import scipy.optimize
import scipy.signal

from . import mel_oracle as mo

MAX_MEM_BLOCK = 2 ** 8 * 2 ** 10          # librosa.util.MAX_MEM_BLOCK [3P]


# --- librosa.util.nnls [3P] -----------------------------------------------------------------------
def _nnls_obj(x, shape, A, B):
    x = x.reshape(shape)
    diff = np.dot(A, x) - B
    value = 0.5 * np.sum(diff ** 2)
    grad = np.dot(A.T, diff)
    return value, grad.flatten()


def _nnls_lbfgs_block(A, B, x_init=None, **kwargs):
    if x_init is None:
        x_init = np.linalg.lstsq(A, B, rcond=None)[0]
        np.clip(x_init, 0, None, out=x_init)
    kwargs.setdefault('m', A.shape[1])
    bounds = [(0, None)] * x_init.size
    shape = x_init.shape
    x, _obj, _diag = scipy.optimize.fmin_l_bfgs_b(_nnls_obj, x_init, args=(shape, A, B), bounds=bounds, **kwargs)
    return x.reshape(shape)


def nnls(A, B, **kwargs):
    if B.ndim == 1:
        return scipy.optimize.nnls(A, B)[0]
    n_columns = int(MAX_MEM_BLOCK // (A.shape[-1] * A.itemsize))
    if B.shape[-1] <= n_columns:
        return _nnls_lbfgs_block(A, B, **kwargs).astype(A.dtype)
    x = np.linalg.lstsq(A, B, rcond=None)[0].astype(A.dtype)
    np.clip(x, 0, None, out=x)
    x_init = x
    for bl_s in range(0, x.shape[-1], n_columns):
        bl_t = min(bl_s + n_columns, B.shape[-1])
        x[:, bl_s:bl_t] = _nnls_lbfgs_block(A, B[:, bl_s:bl_t], x_init=x_init[:, bl_s:bl_t], **kwargs)
    return x


def mel_to_stft(M, sr=22050, n_fft=1024, power=1.0, fmin=0.0, fmax=None):
    mel_basis = mo.mel_filterbank(sr, n_fft, n_mels=M.shape[0], fmin=fmin, fmax=fmax, dtype=M.dtype)
    inverse = nnls(mel_basis, M)
    return np.power(inverse, 1. / power, out=inverse)


# --- librosa.filters.window_sumsquare / core.istft [3P] ----------------------------------------------
def window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=np.float32):
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=dtype)
    win_sq = scipy.signal.get_window('hann', win_length, fftbins=True) ** 2
    win_sq = mo.pad_center(win_sq, n_fft)
    for i in range(n_frames):
        sample = i * hop_length
        x[sample:min(n, sample + n_fft)] += win_sq[:max(0, min(n_fft, n - sample))]
    return x


def istft(stft_matrix, hop_length, win_length, dtype=np.float32):
    n_fft = 2 * (stft_matrix.shape[0] - 1)
    ifft_window = scipy.signal.get_window('hann', win_length, fftbins=True)
    ifft_window = mo.pad_center(ifft_window, n_fft)[:, np.newaxis]
    n_frames = stft_matrix.shape[1]
    y = np.zeros(n_fft + hop_length * (n_frames - 1), dtype=dtype)
    ytmp = ifft_window * np.fft.irfft(stft_matrix, axis=0)
    for frame in range(n_frames):
        sample = frame * hop_length
        y[sample:(sample + n_fft)] += ytmp[:, frame]
    ifft_window_sum = window_sumsquare(n_frames, hop_length, win_length, n_fft, dtype=dtype)
    approx_nonzero_indices = ifft_window_sum > np.finfo(ifft_window_sum.dtype).tiny
    y[approx_nonzero_indices] /= ifft_window_sum[approx_nonzero_indices]
    return y[int(n_fft // 2):-int(n_fft // 2)]


# --- librosa.core.griffinlim [3P] ---------------------------------------------------------------------
def random_phases(shape, random_state):
    """The start phases: exp(2j pi rng.rand(*S.shape)) stored as complex64 (random_state: int seed or RandomState)."""
    rng = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(seed=random_state)
    angles = np.empty(shape, dtype=np.complex64)
    angles[:] = np.exp(2j * np.pi * rng.rand(*shape))
    return angles


def griffinlim(S, n_iter=32, hop_length=256, win_length=1024, momentum=0.99, random_state=0, angles=None,
               return_angles=False):
    n_fft = 2 * (S.shape[0] - 1)
    if angles is None:
        angles = random_phases(S.shape, random_state)
    else:
        angles = np.array(angles, dtype=np.complex64)
    rebuilt = 0.
    for _ in range(n_iter):
        tprev = rebuilt
        inverse = istft(S * angles, hop_length, win_length)
        rebuilt = mo.stft(inverse, n_fft=n_fft, hop_length=hop_length, win_length=win_length)
        angles[:] = rebuilt - (momentum / (1 + momentum)) * tprev
        angles[:] /= np.abs(angles) + 1e-16
    wav = istft(S * angles, hop_length, win_length)
    return (wav, angles) if return_angles else wav


def reconstruct_waveform(mel, normalizer='MelGAN', n_iter=32, sampling_rate=22050, n_fft=1024, hop_length=256,
                         win_length=1024, f_min=0, f_max=8000, random_state=0):
    """Audio.reconstruct_waveform (data/audio.py:94-110) on a normalised mel [n_mels, T]."""
    if normalizer == 'MelGAN':
        amp_mel = np.exp(mel)                                              # data/audio.py:218-219
    else:
        amp_mel = mo.wavernn_denormalize(mel)
    S = mel_to_stft(amp_mel, sr=sampling_rate, n_fft=n_fft, power=1, fmin=f_min, fmax=f_max)
    return griffinlim(S, n_iter=n_iter, hop_length=hop_length, win_length=win_length, random_state=random_state)
