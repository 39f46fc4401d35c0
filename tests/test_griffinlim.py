"""Griffin-Lim (data/audio.py:94-110).

CPU part: the oracle restatement of librosa 0.7.1's mel_to_stft / istft / griffinlim against anchors that need no
librosa (the package is not installable here: parity unpinned, see oracle/griffinlim_oracle.py) and the product's
host-side NNLS against the oracle's.  GPU part: ttsmi_griffinlim against the oracle with the start phases fixed."""
import numpy as np
import pytest
import scipy.signal
import torch

from oracle import griffinlim_oracle as go
from oracle import mel_oracle as mo

HOP, WIN, NFFT, SR = 256, 1024, 1024, 22050


def _speechlike(n, seed):
    r = np.random.RandomState(seed)
    t = np.arange(n) / SR
    f0 = 120 + 30 * np.sin(2 * np.pi * 1.3 * t)
    y = sum(np.sin(2 * np.pi * np.cumsum(f0 * k) / SR + r.rand() * 6) / k for k in range(1, 12))
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 2.1 * t) ** 2
    return (0.1 * env * y + 0.003 * r.randn(n)).astype(np.float32)


def test_istft_inverts_stft_and_matches_scipy():
    y = _speechlike(HOP * 40, 1)
    D = mo.stft(y, NFFT, HOP, WIN)
    back = go.istft(D, HOP, WIN)
    assert back.dtype == np.float32 and back.shape == (HOP * (D.shape[1] - 1),)
    np.testing.assert_allclose(back, y[:len(back)], atol=2e-6)
    # scipy's inverse on the same frames (its scaling convention differs by the window sum)
    _, ys = scipy.signal.istft(D * (1.0 / scipy.signal.get_window('hann', WIN).sum()), fs=1.0, window='hann',
                               nperseg=WIN, noverlap=WIN - HOP, nfft=NFFT, boundary=True)
    np.testing.assert_allclose(ys[:len(back)], back, atol=5e-6)


def test_window_sumsquare_of_periodic_hann_is_the_nola_constant():
    w = go.window_sumsquare(20, HOP, WIN, NFFT)
    assert w.dtype == np.float32 and len(w) == NFFT + HOP * 19
    np.testing.assert_allclose(w[NFFT:-NFFT], 1.5, rtol=1e-6)           # hop = n_fft / 4: sum of hann^2 = 3/2


def test_nnls_solution_satisfies_the_kkt_conditions_and_product_host_code_agrees():
    from transformertts_amd.data import audio as pa
    r = np.random.RandomState(3)
    S_true = (r.rand(513, 140) ** 4).astype(np.float32)
    B = mo.mel_filterbank(SR, NFFT, 80, 0, 8000)
    np.testing.assert_array_equal(pa.mel_filterbank_dense(SR, NFFT, 80, 0, 8000), B)
    M = B @ S_true
    x = go.mel_to_stft(M.copy(), sr=SR, n_fft=NFFT, power=1, fmin=0, fmax=8000)
    assert x.shape == (513, 140) and x.dtype == np.float32 and (x >= 0).all()
    resid = B.astype(np.float64) @ x - M
    grad = B.T.astype(np.float64) @ resid
    # L-BFGS-B stops at a projected gradient of 1e-5 (scipy's default pgtol): that is the optimality it certifies
    assert np.abs(grad[x > 1e-6]).max() < 2e-5                          # stationarity on the free set
    assert grad[x <= 1e-6].min() > -2e-5                                # dual feasibility on the active set
    assert np.linalg.norm(resid) < 1e-2 * np.linalg.norm(M)
    # two blocks (127 + 13 columns) in both implementations; same start point, same optimiser
    xp = pa.mel_to_stft(M.copy(), SR, NFFT, 0, 8000, power=1)
    np.testing.assert_allclose(xp, x, atol=1e-5 * x.max())


def test_griffinlim_oracle_is_seeded_and_converges_on_a_consistent_spectrogram():
    y = _speechlike(HOP * 60, 2)
    S = np.abs(mo.stft(y, NFFT, HOP, WIN))
    a = go.griffinlim(S, n_iter=32, hop_length=HOP, win_length=WIN, random_state=7)
    b = go.griffinlim(S, n_iter=32, hop_length=HOP, win_length=WIN, random_state=7)
    c = go.griffinlim(S, n_iter=32, hop_length=HOP, win_length=WIN, random_state=8)
    np.testing.assert_array_equal(a, b)
    assert np.abs(a - c).max() > 1e-3
    sc = lambda w: np.linalg.norm(np.abs(mo.stft(w, NFFT, HOP, WIN)) - S) / np.linalg.norm(S)
    first = go.griffinlim(S, n_iter=0, hop_length=HOP, win_length=WIN, random_state=7)
    assert sc(a) < 0.25 * sc(first) and sc(a) < 0.2                    # spectral convergence improves a lot


@pytest.mark.gpu
@pytest.mark.parametrize('T,n_iter', [(61, 0), (61, 1), (61, 4), (200, 32), (9, 3)])
def test_gpu_griffinlim_matches_the_oracle_with_fixed_phases(T, n_iter):
    from transformertts_amd.data.audio import Audio
    au = Audio(sampling_rate=SR, n_fft=NFFT, mel_channels=80, hop_length=HOP, win_length=WIN, f_min=0, f_max=8000,
               normalizer='MelGAN')
    y = _speechlike(HOP * (T - 1), 5)
    S = np.abs(mo.stft(y, NFFT, HOP, WIN))
    assert S.shape == (513, T)
    ang0 = go.random_phases(S.shape, 11)
    want, ang_w = go.griffinlim(S, n_iter=n_iter, hop_length=HOP, win_length=WIN, angles=ang0, return_angles=True)
    got, ang_g = au.griffinlim(S, ang0, n_iter=n_iter, return_angles=True)
    assert got.dtype == np.float32 and got.shape == want.shape
    scale = np.abs(want).max()
    # fp32 transforms against NumPy's fp64 ones; the phase normalisation amplifies rounding where |rebuilt| is tiny, so
    # the bound loosens with the iteration count
    # (measured, max error over max sample: 2.4e-7 / 5.0e-7 / 2.2e-6 / 4.2e-6 after 0 / 1 / 3 / 4 iterations, 2.3e-4 after
    # 32 at T = 200 and 5.8e-3 at T = 900 - tools/probe_griffinlim.py)
    tol = {0: 1e-6, 1: 2e-6, 3: 1e-5, 4: 1e-5, 32: 1e-3}[n_iter]
    assert np.abs(got - want).max() < tol * scale, np.abs(got - want).max() / scale
    if 0 < n_iter <= 4:
        big = np.abs(S) > 1e-3 * S.max()                               # phases of bins that carry energy
        assert np.abs(ang_g - ang_w)[big].max() < 1e-2
    sc = lambda w: np.linalg.norm(np.abs(mo.stft(w, NFFT, HOP, WIN)) - S) / np.linalg.norm(S)
    if n_iter == 32:
        assert abs(sc(got) - sc(want)) < 1e-3 and sc(got) < 0.2


@pytest.mark.gpu
def test_reconstruct_waveform_from_a_mel_of_the_model_shape():
    """Audio.reconstruct_waveform end to end against the oracle's, on librosa's own NNLS trajectory (nnls='lbfgs': host
    L-BFGS-B + GPU loop); and the mel of the reconstruction is close to the mel it came from (what the TensorBoard audio
    of the reference is for) - on that path and on the default one (NNLS on the GPU)."""
    from transformertts_amd.data.audio import Audio
    au = Audio(sampling_rate=SR, n_fft=NFFT, mel_channels=80, hop_length=HOP, win_length=WIN, f_min=0, f_max=8000,
               normalizer='MelGAN')
    y = _speechlike(HOP * 150, 9)
    mel = au.mel_spectrogram(y)                                          # [T, 80] normalised
    wav = au.reconstruct_waveform(mel.T, n_iter=32, random_state=4, nnls='lbfgs')
    want = go.reconstruct_waveform(mel.T, 'MelGAN', n_iter=32, random_state=4)
    assert wav.shape == want.shape == (HOP * (mel.shape[0] - 1),)
    assert np.abs(wav - want).max() < 2e-2 * np.abs(want).max()
    loud = mel > mel.max() - 6.0                                         # log-mel bins within e^-6 of the peak
    dev = au.reconstruct_waveform(mel.T, n_iter=32, random_state=4)      # the default: ttsmi_mel_nnls
    assert dev.shape == want.shape and dev.dtype == np.float32
    for w in (wav, dev):
        mel2 = au.mel_spectrogram(np.concatenate([w, np.zeros(HOP, np.float32)]))[:mel.shape[0]]
        assert np.abs(mel2 - mel)[loud].mean() < 0.35
    with pytest.raises(ValueError):
        au.reconstruct_waveform(mel.T, nnls='scipy')
    with pytest.raises(Exception):
        au.griffinlim(np.ones((513, 2), np.float32), np.ones((513, 2), np.complex64))     # too few frames


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['consistent', 'perturbed', 'model-like'])
def test_device_nnls_against_the_oracles_scipy_solution(case):
    """ttsmi_mel_nnls (one wave per frame, accelerated projected gradient) against librosa's L-BFGS-B as the oracle
    restates it: same problem, same start point.  The minimiser is unique only in B x, so the comparison is the
    objective (never worse than scipy's), the KKT conditions scipy certifies, and the distance between the two x."""
    from transformertts_amd.data.audio import Audio
    au = Audio(sampling_rate=SR, n_fft=NFFT, mel_channels=80, hop_length=HOP, win_length=WIN, f_min=0, f_max=8000,
               normalizer='MelGAN')
    r = np.random.RandomState(3)
    B = mo.mel_filterbank(SR, NFFT, 80, 0, 8000).astype(np.float64)
    if case == 'model-like':
        S_true = np.abs(mo.stft(_speechlike(HOP * 139, 6), NFFT, HOP, WIN)).astype(np.float32)
        M = np.exp(np.log(np.maximum(B @ S_true, 1e-5)) + 0.3 * r.randn(80, 140)).astype(np.float32)
    else:
        S_true = (r.rand(513, 140) ** 4).astype(np.float32)
        M = (B @ S_true).astype(np.float32)
        if case == 'perturbed':
            M = (M * np.exp(0.3 * r.randn(*M.shape))).astype(np.float32)
    want = go.mel_to_stft(M.copy(), sr=SR, n_fft=NFFT, power=1, fmin=0, fmax=8000).astype(np.float64)
    x_dev = au.mel_to_stft(M)
    assert tuple(x_dev.shape) == (140, 513) and x_dev.dtype == torch.float32 and x_dev.is_cuda
    got = x_dev.cpu().numpy().T.astype(np.float64)
    assert (got >= 0).all() and np.isfinite(got).all()
    f = lambda x: 0.5 * np.sum((B @ x - M) ** 2)
    assert f(got) <= f(want) * (1 + 1e-3) + 1e-10 * np.sum(M.astype(np.float64) ** 2), (f(got), f(want))
    grad = B.T @ (B @ got - M)
    assert np.abs(grad[got > 1e-6]).max() < 2e-5                        # the bounds the oracle's solution is held to
    assert grad[got <= 1e-6].min() > -2e-5
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 0.2, rel                                               # measured 0.03 (consistent) - 0.09 (perturbed)
    assert torch.equal(au.mel_to_stft(M), x_dev)                        # fixed summation order: bit-reproducible
    np.testing.assert_allclose(au.mel_to_stft(M, power=2.0).cpu().numpy(), np.sqrt(x_dev.cpu().numpy()), rtol=2e-6,
                               atol=1e-12)
    # more steps do not move a converged answer; fewer leave it above scipy's objective at most by a little
    assert f(au.mel_to_stft(M, n_iter=2048).cpu().numpy().T.astype(np.float64)) <= f(got) * (1 + 1e-3) + 1e-12
    assert tuple(au.mel_to_stft(M[:, :0]).shape) == (0, 513)


@pytest.mark.gpu
def test_reconstruct_waveform_of_a_900_frame_mel_takes_milliseconds():
    import time
    from transformertts_amd.data.audio import Audio
    au = Audio(sampling_rate=SR, n_fft=NFFT, mel_channels=80, hop_length=HOP, win_length=WIN, f_min=0, f_max=8000,
               normalizer='MelGAN')
    mel = au.mel_spectrogram(_speechlike(HOP * 899, 12))
    assert mel.shape == (900, 80)
    au.reconstruct_waveform(mel.T, random_state=1)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        wav = au.reconstruct_waveform(mel.T, random_state=1)
        ts.append(time.perf_counter() - t0)
    m = torch.from_numpy(au._denormalize(mel.T)).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    au.mel_to_stft(m)
    torch.cuda.synchronize()
    t_nnls = time.perf_counter() - t0
    print(f'reconstruct_waveform, 900 frames: {min(ts) * 1e3:.1f} ms (mel -> linear on the GPU: {t_nnls * 1e3:.2f} ms)')
    assert wav.shape == (HOP * 899,) and np.isfinite(wav).all()
    # (wall-clock bounds with a wide margin - a busy host stretched one run of 12 to above the former 0.25 s / 20 ms; typical:
    # 60 ms / 3 ms; the host L-BFGS-B path takes tens of seconds here)
    assert min(ts) < 2.0 and t_nnls < 0.25


@pytest.mark.gpu
def test_device_nnls_with_a_basis_whose_bins_are_covered_by_more_than_two_filters():
    """ttsmi_mel_nnls keeps the first two covering filters of a bin in registers and walks the rest of the range in a
    slower loop: a synthetic basis with four overlapping rows per bin (and one empty row, one uncovered bin), against
    scipy.optimize.nnls column by column - the objective it reaches, the KKT conditions, exact zeros where no filter
    reaches."""
    import scipy.optimize
    from transformertts_amd import ops
    r = np.random.RandomState(5)
    n_mels, n_bins, T = 24, 170, 37
    B = np.zeros((n_mels, n_bins), np.float32)
    for j in range(n_mels):
        if j == 5:
            continue                                                    # an empty row
        lo = 6 * j
        B[j, lo:lo + 23] = r.rand(23).astype(np.float32) + 0.1          # rows 6 apart, 23 wide: up to 4 rows per bin
    B[:, 165:] = 0                                                      # bins no filter reaches
    lo = np.array([int(np.nonzero(B[j])[0][0]) if B[j].any() else 0 for j in range(n_mels)], np.int32)
    cnt = np.array([int(np.nonzero(B[j])[0][-1]) + 1 - lo[j] if B[j].any() else 0 for j in range(n_mels)], np.int32)
    ptr = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    w = np.concatenate([B[j, lo[j]:lo[j] + cnt[j]] for j in range(n_mels)]).astype(np.float32)
    assert ((B > 0).sum(0).max()) == 4
    M = (B @ (r.rand(n_bins, T) ** 3) * np.exp(0.3 * r.randn(n_mels, T))).astype(np.float32)
    B64 = B.astype(np.float64)
    dev = torch.device('cuda:0')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = ops.mel_nnls(t(M.T), t(np.linalg.pinv(B64).T.astype(np.float32)), t(lo), t(cnt), t(ptr), t(w),
                     1.0 / np.linalg.norm(B64, 2) ** 2, n_iter=4000)
    got = x.cpu().numpy().T.astype(np.float64)
    assert got.shape == (n_bins, T) and (got >= 0).all() and (got[165:] == 0).all()
    f = lambda X: 0.5 * np.sum((B64 @ X - M) ** 2)
    want = np.stack([scipy.optimize.nnls(B64, M[:, c].astype(np.float64))[0] for c in range(T)], 1)
    assert f(got) <= f(want) * (1 + 1e-3) + 1e-9 * np.sum(M.astype(np.float64) ** 2), (f(got), f(want))
    grad = B64.T @ (B64 @ got - M)
    scale = np.abs(B64.T @ M).max()
    assert np.abs(grad[got > 1e-5 * got.max()]).max() < 1e-4 * scale and grad.min() > -1e-4 * scale
