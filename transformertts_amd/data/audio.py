"""Audio - mirror of the mel-extraction path of reference data/audio.py (class Audio :13-92,196-198,
normalisers :201-242) on the batched HIP STFT->mel kernel (ttsmi_stft_logmel).

`Audio.from_config(cfg).mel_spectrogram(wav)` keeps the reference signature (float32 wav [N] ->
float32 [frames, mel_channels]); `mel_spectrogram_batch` is the MI355X-shaped entry: many clips in
one launch, concatenated with offset tables, device-resident in and out.

Host-side constants (built once, in float64 numpy like librosa does [3P], then cast): the periodic
Hann window centred in n_fft, and the Slaney/area-normalised mel filterbank in its sparse
per-filter (first bin, count, weights) form.  `preprocess` keeps the two steps of the reference's
wav preparation that decide what the mel kernel sees (data/audio.py:132-141):
volume normalisation and the one-sample pad that fixes the frame count.  Everything else of the
reference class (wav loading, VAD / silence trimming, pyworld pitch, plotting: data/audio.py:112-194) is
outside the hot path and not provided; asking for the trimming steps raises
instead of silently skipping them.

`reconstruct_waveform` (data/audio.py:94-110) is split where the reference's own cost
splits: the mel -> linear-magnitude non-negative least squares is a one-off host computation (librosa runs it
through scipy's L-BFGS-B; so does `mel_to_stft` below, restating librosa.util.nnls 0.7.1 [3P]), the Griffin-Lim
loop - 33 inverse and 32 forward short-time transforms - runs on the GPU (ttsmi_griffinlim)."""
from __future__ import annotations

import sys
from typing import List, Sequence

import numpy as np
import torch

from .. import ops


# ---- Slaney mel scale (librosa.filters.mel(htk=False, norm=1)) [3P] ------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    lin = f / f_sp
    with np.errstate(divide='ignore', invalid='ignore'):
        log = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank_sparse(sr, n_fft, n_mels, fmin, fmax):
    """Returns (lo [n_mels] i32, cnt [n_mels] i32, ptr [n_mels] i32, weights f32): filter m covers
    FFT bins lo[m] .. lo[m]+cnt[m] with weights[ptr[m] ..]."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(float(fmin)), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lo, cnt, ptr, ws = [], [], [], []
    for i in range(n_mels):
        tri = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
        # float32 triangle scaled by the float64 area norm, rounded once - as numpy's in-place
        # `weights *= enorm[:, None]` on a float32 array does
        tri = (tri.astype(np.float32).astype(np.float64) * (2.0 / (mel_f[i + 2] - mel_f[i]))).astype(np.float32)
        nz = np.nonzero(tri)[0]
        if len(nz) == 0:
            lo.append(0); cnt.append(0); ptr.append(len(ws)); continue
        a, b = int(nz[0]), int(nz[-1]) + 1
        lo.append(a); cnt.append(b - a); ptr.append(len(ws))
        ws.extend(tri[a:b].tolist())
    return (np.asarray(lo, np.int32), np.asarray(cnt, np.int32), np.asarray(ptr, np.int32),
            np.asarray(ws, np.float32))


def hann_window_padded(win_length: int, n_fft: int) -> np.ndarray:
    """scipy.signal.get_window('hann', win_length, fftbins=True) centred in n_fft [3P]."""
    n = np.arange(win_length, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_length)
    lpad = (n_fft - win_length) // 2
    return np.pad(w, (lpad, n_fft - win_length - lpad)).astype(np.float32)


# ---- librosa.feature.inverse.mel_to_stft / librosa.util.nnls (0.7.1) [3P], host side ---------------------
_NNLS_BLOCK_BYTES = 2 ** 8 * 2 ** 10          # librosa.util.MAX_MEM_BLOCK


def mel_filterbank_dense(sr, n_fft, n_mels, fmin, fmax, dtype=np.float32) -> np.ndarray:
    """[n_mels, 1 + n_fft // 2] Slaney / area-normalised basis (the dense form of mel_filterbank_sparse)."""
    lo, cnt, ptr, w = mel_filterbank_sparse(sr, n_fft, n_mels, fmin, fmax)
    B = np.zeros((n_mels, 1 + n_fft // 2), dtype=dtype)
    for m in range(n_mels):
        B[m, lo[m]:lo[m] + cnt[m]] = w[ptr[m]:ptr[m] + cnt[m]]
    return B


def _nnls_block(A, B, x0):
    import scipy.optimize

    def obj(x):
        X = x.reshape(x0.shape)
        diff = A @ X - B
        return 0.5 * float(np.sum(diff * diff)), (A.T @ diff).ravel()
    x, _, _ = scipy.optimize.fmin_l_bfgs_b(obj, x0, bounds=[(0, None)] * x0.size, m=A.shape[1])
    return x.reshape(x0.shape)


def mel_to_stft(M: np.ndarray, sr: int, n_fft: int, fmin: float, fmax: float, power: float = 1.0) -> np.ndarray:
    """Linear magnitudes [1 + n_fft // 2, T] whose mel projection approximates M [n_mels, T]: bounded L-BFGS from the
    clipped least-squares solution, in column blocks of MAX_MEM_BLOCK bytes, as librosa.util.nnls does."""
    M = np.ascontiguousarray(M)
    A = mel_filterbank_dense(sr, n_fft, M.shape[0], fmin, fmax, dtype=M.dtype)
    x = np.linalg.lstsq(A, M, rcond=None)[0]
    np.clip(x, 0, None, out=x)
    ncol = int(_NNLS_BLOCK_BYTES // (A.shape[-1] * A.itemsize))
    if M.shape[-1] <= ncol:
        x = _nnls_block(A, M, x).astype(A.dtype)
    else:
        x = x.astype(A.dtype)
        for s0 in range(0, x.shape[-1], ncol):
            s1 = min(s0 + ncol, M.shape[-1])
            x[:, s0:s1] = _nnls_block(A, M[:, s0:s1], x[:, s0:s1])
    return np.power(x, 1.0 / power, out=x)


def window_sumsquare(window: np.ndarray, n_frames: int, hop_length: int) -> np.ndarray:
    """librosa.filters.window_sumsquare [3P]: overlap-add envelope of window**2 (fp32 accumulation, frame order)."""
    n_fft = len(window)
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    win_sq = window.astype(np.float64) ** 2
    for i in range(n_frames):
        s0 = i * hop_length
        x[s0:min(n, s0 + n_fft)] += win_sq[:max(0, min(n_fft, n - s0))]
    return x


class Normalizer:
    def normalize(self, S):
        raise NotImplementedError

    def denormalize(self, S):
        raise NotImplementedError


class MelGAN(Normalizer):                       # reference data/audio.py:209-219
    kernel_id = 0

    def __init__(self):
        self.clip_min = 1.0e-5

    def denormalize(self, S):
        return torch.exp(S) if torch.is_tensor(S) else np.exp(S)


class WaveRNN(Normalizer):                      # reference data/audio.py:222-242
    kernel_id = 1

    def __init__(self):
        self.min_level_db = -100
        self.max_norm = 4
        self.clip_min = 1.0e-5

    def denormalize(self, S):
        S = (S + self.max_norm) / (2 * self.max_norm)
        S = (np.clip(S, 0, 1) * -self.min_level_db) + self.min_level_db
        return np.power(10.0, S * 0.05)


def normalize_volume(wav: np.ndarray, target_dBFS: float, int16_max: float, increase_only: bool = False,
                     decrease_only: bool = False) -> np.ndarray:
    """Reference Audio.normalize_volume (data/audio.py:154-162): scale to `target_dBFS` RMS."""
    if increase_only and decrease_only:
        raise ValueError('Both increase only and decrease only are set')
    rms = np.sqrt(np.mean((wav * int16_max) ** 2))
    wave_dBFS = 20 * np.log10(rms / int16_max)
    dBFS_change = target_dBFS - wave_dBFS
    if (dBFS_change < 0 and increase_only) or (dBFS_change > 0 and decrease_only):
        return wav
    return wav * (10 ** (dBFS_change / 20))


def pad_for_frame_count(y: np.ndarray, hop_length: int) -> np.ndarray:
    """Reference Audio.preprocess, last step (data/audio.py:139-140): a clip whose length is a multiple
    of the hop gets ONE zero sample appended, so every clip handed to the STFT has len % hop != 0 and its
    frame count 1 + len // hop is the one the alignment / duration extraction was computed against."""
    if y.shape[0] % hop_length == 0:
        y = np.pad(y, (0, 1))
    return y


class Audio:
    def __init__(self, sampling_rate: int, n_fft: int, mel_channels: int, hop_length: int,
                 win_length: int, f_min: int, f_max: int, normalizer: str, norm_wav: bool = None,
                 target_dBFS: int = None, int16_max: int = None, trim_long_silences: bool = None,
                 trim_silence: bool = None, trim_silence_top_db: int = None,
                 vad_window_length: int = None, vad_sample_rate: int = None,
                 vad_moving_average_width: int = None, vad_max_silence_length: int = None, **kwargs):
        self.config = {k: v for k, v in locals().items() if k not in ('self', '__class__', 'kwargs')}
        self.sampling_rate, self.n_fft, self.mel_channels = sampling_rate, n_fft, mel_channels
        self.hop_length, self.win_length, self.f_min, self.f_max = hop_length, win_length, f_min, f_max
        self.norm_wav, self.target_dBFS, self.int16_max = norm_wav, target_dBFS, int16_max
        self.trim_long_silences, self.trim_silence = trim_long_silences, trim_silence
        self.normalizer = getattr(sys.modules[__name__], normalizer)()
        self.device = torch.device(kwargs.get('device', 'cuda:0'))
        if not torch.cuda.is_available():
            raise ops._lib.TtsmiError('Audio.mel_spectrogram runs on an MI355X through libttsmi.so; no GPU '
                                      'is visible and there is no CPU fallback')
        ops._lib.lib()
        lo, cnt, ptr, w = mel_filterbank_sparse(sampling_rate, n_fft, mel_channels, f_min, f_max)
        dev = self.device
        self._mel = tuple(torch.from_numpy(a).to(dev) for a in (lo, cnt, ptr, w))
        self._window_host = hann_window_padded(win_length, n_fft)
        self._window = torch.from_numpy(self._window_host).to(dev)
        self._nnls = None                         # (pinv(B)^T on the device, 1 / ||B||_2^2): built on first use

    @classmethod
    def from_config(cls, config: dict):
        return cls(**config)

    def mel_spectrogram_batch(self, wavs: Sequence, lengths: Sequence[int] = None):
        """wavs: list of 1-D float32 arrays/tensors, OR one already-concatenated device tensor with
        `lengths`.  Returns (mel [total_frames, mel_channels] on the device, frame_off int64 [n+1])."""
        if torch.is_tensor(wavs) and lengths is not None:
            cat = wavs.to(self.device, torch.float32).contiguous()
            lengths = [int(x) for x in lengths]
        else:
            lengths = [int(len(w)) for w in wavs]
            cat = torch.cat([torch.as_tensor(np.asarray(w.cpu() if torch.is_tensor(w) else w),
                                             dtype=torch.float32) for w in wavs]).to(self.device)
        if min(lengths) <= self.n_fft // 2:
            raise ValueError(f'clips must be longer than n_fft/2 = {self.n_fft // 2} samples (reflect padding)')
        clip_off = np.zeros(len(lengths) + 1, dtype=np.int64)
        clip_off[1:] = np.cumsum(lengths)
        frames = [1 + n // self.hop_length for n in lengths]          # librosa center=True
        frame_off = np.zeros(len(lengths) + 1, dtype=np.int64)
        frame_off[1:] = np.cumsum(frames)
        lo, cnt, ptr, w = self._mel
        with torch.cuda.device(self.device):      # the launch stream follows torch's CURRENT device
            out = ops.stft_logmel(cat, torch.from_numpy(clip_off).to(self.device),
                                  torch.from_numpy(frame_off).to(self.device), int(frame_off[-1]), self.n_fft,
                                  self.hop_length, self._window, self.mel_channels, lo, cnt, ptr, w,
                                  self.normalizer.kernel_id, float(self.normalizer.clip_min))
        return out, frame_off

    def mel_spectrogram(self, wav):
        """Reference Audio.mel_spectrogram (data/audio.py:88-92): wav [N] -> [frames, mel_channels].
        numpy in -> numpy out; CUDA tensor in -> CUDA tensor out."""
        is_t = torch.is_tensor(wav)
        mel, _ = self.mel_spectrogram_batch([wav])
        return mel if is_t else mel.cpu().numpy()

    def normalize_volume(self, wav, increase_only=False, decrease_only=False):
        return normalize_volume(wav, self.target_dBFS, self.int16_max, increase_only, decrease_only)

    def preprocess(self, y):
        """Reference Audio.preprocess (data/audio.py:132-141).  The VAD / silence trimming steps need
        webrtcvad / librosa and are out of scope: configured on, they raise."""
        if self.norm_wav:
            y = self.normalize_volume(y, increase_only=True)
        if self.trim_long_silences or self.trim_silence:
            raise NotImplementedError('trim_long_silences / trim_silence (webrtcvad, librosa.effects.trim) are '
                                      'outside the hot path: trim the clips upstream and configure them off')
        return pad_for_frame_count(y, self.hop_length)

    def mel_to_stft(self, amp_mel, n_iter=512, power=1.0):
        """librosa.feature.inverse.mel_to_stft [3P] on the GPU (ttsmi_mel_nnls): amplitude mel [mel_channels, T] (numpy or
        tensor) -> DEVICE tensor [T, 1 + n_fft // 2] of non-negative linear magnitudes, frame-major.  Same problem and
        start point as the host function of this module (scipy L-BFGS-B, librosa's optimiser); the minimiser is unique
        in its mel projection, not in x, and this one reaches a lower objective than scipy's stopping rule does."""
        dev = self.device
        if self._nnls is None:
            B = mel_filterbank_dense(self.sampling_rate, self.n_fft, self.mel_channels, self.f_min, self.f_max,
                                     dtype=np.float64)
            pinv_t = np.ascontiguousarray(np.linalg.pinv(B).T, dtype=np.float32)          # [n_mels, n_bins]
            self._nnls = (torch.from_numpy(pinv_t).to(dev), float(1.0 / np.linalg.norm(B, 2) ** 2))
        pinv_t, inv_l = self._nnls
        m = amp_mel if torch.is_tensor(amp_mel) else torch.from_numpy(np.asarray(amp_mel))
        m = m.detach().to(dev, torch.float32).t().contiguous()                            # [T, n_mels]
        lo, cnt, ptr, w = self._mel
        with torch.cuda.device(dev):
            return ops.mel_nnls(m, pinv_t, lo, cnt, ptr, w, inv_l, n_iter=n_iter, power=power)

    def reconstruct_waveform(self, mel, n_iter=32, random_state=None, momentum=0.99, nnls='device', nnls_iter=512):
        """Reference Audio.reconstruct_waveform (data/audio.py:94-110): normalised mel [mel_channels, T] (callers pass
        `mel.T`, predict_tts.py:56) -> float32 wav [hop * (T - 1)].  `random_state` (int seed / RandomState / None)
        draws the start phases exactly as librosa.griffinlim does (None = NumPy's global generator: unseeded, like the
        reference's call).  nnls: 'device' (ttsmi_mel_nnls, milliseconds) or 'lbfgs' (librosa's own optimiser on the
        host, seconds to minutes: the trajectory the reference takes - the two agree in the mel projection of the
        result, not bin by bin, see mel_to_stft)."""
        mel = np.asarray(mel.detach().cpu() if torch.is_tensor(mel) else mel)
        amp_mel = self._denormalize(mel)
        if nnls == 'device':
            S = self.mel_to_stft(amp_mel, n_iter=nnls_iter)                                # [T, bins] on the device
            shape = (int(S.shape[1]), int(S.shape[0]))
        elif nnls == 'lbfgs':
            S = mel_to_stft(amp_mel, self.sampling_rate, self.n_fft, self.f_min, self.f_max, power=1)
            shape = S.shape
        else:
            raise ValueError(f"nnls must be 'device' or 'lbfgs', not {nnls!r}")
        rng = (np.random if random_state is None else random_state if isinstance(random_state, np.random.RandomState)
               else np.random.RandomState(seed=random_state))
        angles = np.empty(shape, dtype=np.complex64)
        angles[:] = np.exp(2j * np.pi * rng.rand(*shape))
        return self.griffinlim(S, angles, n_iter=n_iter, momentum=momentum)

    def griffinlim(self, S, angles, n_iter=32, momentum=0.99, return_angles=False):
        """librosa.core.griffinlim [3P] on the GPU: S [1 + n_fft // 2, T] linear magnitudes (numpy), or a DEVICE tensor
        [T, 1 + n_fft // 2] (frame-major, as mel_to_stft returns it); `angles` the complex start phases [bins, T]."""
        dev = self.device
        T = int(S.shape[0] if torch.is_tensor(S) else S.shape[1])
        with torch.cuda.device(dev):
            mag = (S.to(dev, torch.float32).contiguous() if torch.is_tensor(S)
                   else torch.from_numpy(np.ascontiguousarray(S.T, dtype=np.float32)).to(dev))
            a = np.ascontiguousarray(np.asarray(angles, dtype=np.complex64).T)
            ang = torch.from_numpy(a.view(np.float32).reshape(T, -1, 2).copy()).to(dev)
            wss = torch.from_numpy(window_sumsquare(self._window_host, T, self.hop_length)).to(dev)
            wav = ops.griffinlim(mag, ang, self._window, wss, self.n_fft, self.hop_length, n_iter, momentum)
            out = wav.cpu().numpy()
            if return_angles:
                return out, np.ascontiguousarray(ang.cpu().numpy()).view(np.complex64).reshape(T, -1).T
            return out

    def _normalize(self, S):
        raise NotImplementedError('normalisation is fused into the STFT->mel kernel')

    def _denormalize(self, S):
        return self.normalizer.denormalize(S)
