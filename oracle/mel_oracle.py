"""CPU restatement ORACLE of the wav -> STFT -> mel -> log path (data/audio.py:72-92,209-242).

TEST INFRASTRUCTURE ONLY (see oracle/ft_oracle.py header).  PARITY UNPINNED: the arithmetic lives
in librosa==0.7.1 (requirements.txt:2, with numba==0.48 / numpy>=1.17.4), which is not installed
and not vendored under /root/reference.  Its published algorithm is restated below [3P]:

  librosa.stft(y, n_fft, hop_length, win_length)            (call site data/audio.py:81-86)
      window = scipy.signal.get_window('hann', win_length, fftbins=True)   (periodic Hann, fp64)
      window = pad_center(window, n_fft)
      y      = np.pad(y, n_fft // 2, mode='reflect')                       (center=True)
      frames = y[t*hop : t*hop + n_fft],  t = 0 .. (len(y_padded) - n_fft) // hop
      D      = rfft(window * frames)  computed in fp64, stored as complex64
  librosa.feature.melspectrogram(S=|D|, sr, n_fft, n_mels, fmin, fmax) (call site audio.py:72-79)
      S given  =>  used as is (magnitude, i.e. power=1 semantics), mel_basis (fp32) @ S (fp32)
  librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm=1)   Slaney scale + area norm

Pinned against the reference's own code (tests/golden/make_reference_fixtures.py ->
tests/test_reference_fixtures.py): the MelGAN / WaveRNN normalisers (data/audio.py:209-242, plain NumPy in the
reference) - the STFT and the mel basis (librosa) are not.
Anchors available without librosa (tests/test_oracle.py): the Slaney mel-frequency table from
librosa's public documentation, torch.stft(center=True, reflect,
periodic hann) on CPU, scipy's get_window, analytic inputs (silence, pure sine), and a third party's NumPy restatement
of librosa that is installed here (transformers.audio_utils: filterbank to rounding, log-mel to 1e-6).
"""
from __future__ import annotations

import numpy as np
import scipy.signal


# --- librosa.core.time_frequency [3P] --------------------------------------------------------
def hz_to_mel(frequencies):
    frequencies = np.asanyarray(frequencies, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    mels = (frequencies - f_min) / f_sp
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if frequencies.ndim:
        log_t = frequencies >= min_log_hz
        mels[log_t] = min_log_mel + np.log(frequencies[log_t] / min_log_hz) / logstep
    elif frequencies >= min_log_hz:
        mels = min_log_mel + np.log(frequencies / min_log_hz) / logstep
    return mels


def mel_to_hz(mels):
    mels = np.asanyarray(mels, dtype=np.float64)
    f_min, f_sp = 0.0, 200.0 / 3
    freqs = f_min + f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = (min_log_hz - f_min) / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        log_t = mels >= min_log_mel
        freqs[log_t] = min_log_hz * np.exp(logstep * (mels[log_t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0):
    min_mel = hz_to_mel(fmin)
    max_mel = hz_to_mel(fmax)
    mels = np.linspace(min_mel, max_mel, n_mels)
    return mel_to_hz(mels)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, dtype=np.float32):
    """librosa.filters.mel(htk=False, norm=1) [3P]."""
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=dtype)
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


# --- librosa.core.spectrum.stft [3P] ---------------------------------------------------------
def pad_center(data, size):
    n = data.shape[-1]
    lpad = int((size - n) // 2)
    return np.pad(data, (lpad, int(size - n - lpad)), mode='constant')


def stft(y, n_fft=1024, hop_length=256, win_length=1024):
    y = np.asarray(y)
    fft_window = scipy.signal.get_window('hann', win_length, fftbins=True)
    fft_window = pad_center(fft_window, n_fft).reshape((-1, 1))
    y = np.pad(y, int(n_fft // 2), mode='reflect')
    n_frames = 1 + (len(y) - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    y_frames = y[idx]                                             # [n_fft, n_frames]
    D = np.fft.rfft(fft_window * y_frames, axis=0)                # fp64 (fp64 window * fp32 frames)
    return D.astype(np.complex64)


# --- data/audio.py normalisers ---------------------------------------------------------------
def melgan_normalize(S, clip_min=1.0e-5):
    """MelGAN.normalize data/audio.py:214-216."""
    return np.log(np.clip(S, a_min=clip_min, a_max=None))


def wavernn_normalize(S, min_level_db=-100, max_norm=4):
    """WaveRNN.normalize data/audio.py:228-239."""
    S = 20 * np.log10(np.maximum(1e-5, S))
    S = np.clip((S - min_level_db) / -min_level_db, 0, 1)
    return (S * 2 * max_norm) - max_norm


def wavernn_denormalize(S, min_level_db=-100, max_norm=4):
    """WaveRNN.denormalize data/audio.py:233-236,241-242."""
    S = (S + max_norm) / (2 * max_norm)
    S = (np.clip(S, 0, 1) * -min_level_db) + min_level_db
    return np.power(10.0, S * 0.05)


def mel_spectrogram(wav, sampling_rate=22050, n_fft=1024, mel_channels=80, hop_length=256,
                    win_length=1024, f_min=0, f_max=8000, normalizer='MelGAN', exact=False):
    """Audio.mel_spectrogram data/audio.py:88-92.  Returns float32 [frames, mel_channels].
    exact=True keeps everything in fp64 (truth for tolerance accounting)."""
    wav = np.asarray(wav, dtype=np.float32)
    D = stft(wav, n_fft, hop_length, win_length)
    basis = mel_filterbank(sampling_rate, n_fft, mel_channels, f_min, f_max)
    if exact:
        y = np.pad(wav.astype(np.float64), n_fft // 2, mode='reflect')
        win = pad_center(scipy.signal.get_window('hann', win_length, fftbins=True), n_fft)
        nfr = 1 + (len(y) - n_fft) // hop_length
        idx = np.arange(n_fft)[:, None] + hop_length * np.arange(nfr)[None, :]
        mag = np.abs(np.fft.rfft(win[:, None] * y[idx], axis=0))
        S = basis.astype(np.float64) @ mag
    else:
        S = np.dot(basis, np.abs(D))                              # fp32 @ fp32
    if normalizer == 'MelGAN':
        out = melgan_normalize(S)
    elif normalizer == 'WaveRNN':
        out = wavernn_normalize(S)
    else:
        raise ValueError(normalizer)
    return out.T


def synthetic_clip(n_samples: int, seed: int) -> np.ndarray:
    """seeded N(0, 0.1^2) noise + 2 sines, float32."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / 22050.0
    f1, f2 = rng.uniform(80, 400), rng.uniform(500, 4000)
    y = 0.1 * rng.standard_normal(n_samples) + 0.3 * np.sin(2 * np.pi * f1 * t) \
        + 0.2 * np.sin(2 * np.pi * f2 * t)
    return y.astype(np.float32)
