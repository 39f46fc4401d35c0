"""Pins the CPU oracle against everything available without TF/librosa:
the Expand docstring example (the only reference-authored KAT on the path), the Slaney
mel-frequency table, torch.stft / scipy conventions, analytic mel inputs, TF-Adam hand values."""
import math

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo
from oracle import mel_oracle as mo


# ---------------------------------------------------------------- Expand (model/layers.py:527-565)
def test_expand_docstring_example():
    # model/layers.py:532-542
    x = np.array([[[0.54710746, 0.8943467], [0.7140938, 0.97968304], [0.5347662, 0.15213418]]],
                 dtype=np.float32)
    dims = np.array([[[1], [3], [2]]], dtype=np.int32)
    want = np.array([[[0.54710746, 0.8943467], [0.7140938, 0.97968304], [0.7140938, 0.97968304],
                      [0.7140938, 0.97968304], [0.5347662, 0.15213418], [0.5347662, 0.15213418]]],
                    dtype=np.float32)
    got = fo.expand_literal_np(x, dims)
    assert got.shape == (1, 6, 2)
    np.testing.assert_array_equal(got, want)
    got_t = fo.expand_torch(torch.from_numpy(x), torch.from_numpy(dims)).numpy()
    np.testing.assert_array_equal(got_t, want)


@pytest.mark.parametrize('seed', range(5))
def test_expand_literal_equals_index_form(seed):
    rng = np.random.default_rng(seed)
    B, T, C = 3, 7, 4
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    dur = rng.choice([0., 0.5, 1., 1.5, 2.5, 3.5, 2.49, 4.0], size=(B, T, 1)).astype(np.float32)
    if seed == 0:
        dur[1] = 0           # an all-zero sample
    lit = fo.expand_literal_np(x, dur)
    idx, lens, out_len = fo.expand_indices_np(dur)
    assert lit.shape[1] == out_len
    ref = np.zeros_like(lit)
    for b in range(B):
        for j in range(lens[b]):
            ref[b, j] = x[b, idx[b, j]]
    np.testing.assert_array_equal(lit, ref)
    np.testing.assert_array_equal(fo.expand_torch(torch.from_numpy(x), torch.from_numpy(dur)).numpy(), lit)


def test_round_half_even_ties():
    np.testing.assert_array_equal(fo.round_half_even(np.array([0.5, 1.5, 2.5, 3.5, -0.5, 2.4999])),
                                  np.array([0., 2., 2., 4., -0., 2.]))


# ---------------------------------------------------------------- positional encoding / masks
def test_positional_encoding_values():
    pe = fo.positional_encoding(50, 16)
    assert pe.dtype == np.float32 and pe.shape == (50, 16)
    np.testing.assert_allclose(pe[0, 0::2], 0.0)
    np.testing.assert_allclose(pe[0, 1::2], 1.0)
    # column i uses exponent 2*(i//2)/d  (transformer_utils.py:5-7)
    for p, i in [(3, 4), (7, 5), (49, 15)]:
        ang = p / (10000 ** (2 * (i // 2) / 16))
        want = math.sin(ang) if i % 2 == 0 else math.cos(ang)
        assert abs(pe[p, i] - want) < 1e-6


def test_padding_masks():
    tok = torch.tensor([[3, 0, 5, 0]])
    m = fo.create_encoder_padding_mask(tok, torch.float32)
    assert m.shape == (1, 1, 1, 4)
    np.testing.assert_array_equal(m[0, 0, 0].numpy(), [0, 1, 0, 1])
    mel = torch.tensor([[[1., -1.], [0., 0.], [0., 1e-9]]])
    np.testing.assert_array_equal(fo.create_mel_padding_mask(mel)[0, 0, 0].numpy(), [0, 1, 0])


# ---------------------------------------------------------------- building blocks vs torch.nn
def test_layer_norm_matches_torch():
    x = torch.randn(5, 7, dtype=torch.float64)
    g, b = torch.randn(7, dtype=torch.float64), torch.randn(7, dtype=torch.float64)
    np.testing.assert_allclose(fo.layer_norm(x, g, b).numpy(),
                               torch.nn.functional.layer_norm(x, (7,), g, b, eps=1e-6).numpy(),
                               rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('k', [3, 5, 4])
def test_conv1d_same_matches_torch(k):
    x = torch.randn(2, 9, 3, dtype=torch.float64)
    w = torch.randn(k, 3, 5, dtype=torch.float64)
    b = torch.randn(5, dtype=torch.float64)
    got = fo.conv1d_same(x, w, b)
    left, right = (k - 1) // 2, k // 2
    xp = torch.nn.functional.pad(x.transpose(1, 2), (left, right))
    want = torch.nn.functional.conv1d(xp, w.permute(2, 1, 0), b).transpose(1, 2)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-12, atol=1e-12)


def test_unmasked_mae_counts_padding():
    # utils/losses.py:41-49: mask stays None => plain mean including padded zeros
    t = torch.tensor([[[1.], [0.]]])
    p = torch.tensor([[[0.5], [0.25]]])
    assert float(fo.masked_mean_absolute_error(t, p)) == pytest.approx((0.5 + 0.25) / 2)
    ti = torch.tensor([[[2], [0]]], dtype=torch.int32)
    assert float(fo.masked_mean_absolute_error(ti, p)) == pytest.approx((1.5 + 0.25) / 2)


def test_keras_loss_reduction_matches_the_reference_known_answers():
    """tests/test_loss.py:10-26 of the reference: the three known answers of its crossentropy losses.  They pin what a
    Keras loss object does with sample weights - the weighted sum is divided by the number of samples, not by the sum
    of the weights - which is the reduction the oracle's MAE uses with no weights."""
    targets = torch.tensor([[0, 1, 2]])
    logits = torch.tensor([[[.3, .2, .1], [.3, .2, .1], [.3, .2, .1]]])
    assert float(fo.masked_crossentropy(targets, logits, index=2, scaling=5)) == pytest.approx(2.3705523014068604, abs=1e-6)
    assert float(fo.masked_crossentropy(targets, logits, index=2, scaling=1)) == pytest.approx(0.7679619193077087, abs=1e-6)
    assert float(fo.masked_crossentropy(targets, logits)) == pytest.approx(0.7679619193077087, abs=1e-6)
    # dividing by the sum of the weights instead would give 1.15194...: the known answers exclude it
    assert abs(0.7679619193077087 * 3 / 2 - 1.1519428789615631) < 1e-12
    x = torch.randn(4, 7, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    assert float(fo.keras_weighted_loss_mean(x.abs().mean(-1))) == pytest.approx(float(x.abs().mean()))


def test_tf_adam_hand_values():
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=0)
    m = fo.ForwardTransformerOracle(cfg, W)
    g = {k: torch.full_like(v, 0.5) for k, v in m.W.items()}
    w0 = m.W['out.b'].detach().clone()
    m.learning_rate = 1e-3
    m.apply_gradients(g)
    # step 1: m=0.05, v=0.005, lr_t = lr*sqrt(1-.98)/(1-.9) ; theta -= lr_t*m/(sqrt(v)+eps)
    lr_t = 1e-3 * math.sqrt(1 - 0.98) / (1 - 0.9)
    want = w0 - lr_t * 0.05 / (math.sqrt(0.005) + 1e-9)
    np.testing.assert_allclose(m.W['out.b'].detach().numpy(), want.numpy(), rtol=1e-12)
    assert m.step == 1


# ---------------------------------------------------------------- whole model, tiny config
def test_tiny_forward_shapes_and_quirks():
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=1, perturb=0.05)
    m = fo.ForwardTransformerOracle(cfg, W)
    tok, mel, dur, pitch = fo.synthetic_batch(3, 12, 40, seed=2, ragged=True)
    out = m.val_step(tok, mel, dur, pitch)
    assert out['mel'].shape == (3, 40, 80)
    assert out['duration'].shape == (3, 12, 1) and out['pitch'].shape == (3, 12, 1)
    assert out['expanded_mask'].shape == (3, 1, 1, 40)
    assert list(out['encoder_attention']) == ['Encoder_DenseBlock1_SelfAttention',
                                              'Encoder_DenseBlock2_SelfAttention']
    assert out['decoder_attention']['Decoder_DenseBlock2_SelfAttention'].shape == (3, 2, 40, 40)
    # padded decoder rows are exactly the output bias (layers.py:230 zeroes them, models.py:543)
    lens = dur.sum(1)
    b = int(np.argmin(lens))
    assert lens[b] < 40
    np.testing.assert_allclose(out['mel'][b, lens[b]:].numpy(),
                               np.broadcast_to(W['out.b'], (40 - lens[b], 80)), atol=1e-12)
    # predictors are masked at padded phonemes (layers.py:485)
    pad = tok == 0
    assert np.all(out['duration'].numpy()[pad] == 0) and np.all(out['pitch'].numpy()[pad] == 0)
    # loss = 1*mel + 1*dur + 3*pitch  (models.py:485)
    l = out['losses']
    assert float(out['loss']) == pytest.approx(float(l['mel'] + l['duration'] + 3 * l['pitch']))


def test_tiny_train_step_reduces_loss_fp32_close_to_fp64():
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=3, perturb=0.02)
    batch = fo.synthetic_batch(2, 10, 30, seed=4)
    m64 = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    m32 = fo.ForwardTransformerOracle(cfg, W, torch.float32)
    m64.learning_rate = m32.learning_rate = 1e-3
    l64 = [float(m64.train_step(*batch)['loss']) for _ in range(5)]
    l32 = [float(m32.train_step(*batch)['loss']) for _ in range(5)]
    assert l64[-1] < l64[0]
    np.testing.assert_allclose(l32, l64, rtol=2e-5)


# ---------------------------------------------------------------- mel path (data/audio.py:72-92)
def test_slaney_mel_frequency_table():
    # librosa documentation example mel_frequencies(n_mels=40) (fmin 0, fmax 11025), SURVEY 8c.3
    f = mo.mel_frequencies(40, 0.0, 11025.0)
    head = [0., 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855,
            853.173, 938.49, 1024.856, 1119.114, 1222.042, 1334.436]
    np.testing.assert_allclose(f[:16], head, atol=5e-4)
    np.testing.assert_allclose(f[-3:], [9246.028, 10096.408, 11025.], atol=5e-4)


def test_lj_filterbank_structure():
    B = mo.mel_filterbank(22050, 1024, 80, 0, 8000)
    assert B.shape == (80, 513) and B.dtype == np.float32
    assert np.count_nonzero(B) == 727
    assert np.nonzero(B.sum(0))[0].max() == 371
    assert abs(B.max() - 0.02649) < 1e-4
    assert (B.sum(1) > 0).all()


def test_stft_matches_torch_stft():
    y = mo.synthetic_clip(5000, seed=0)
    D = mo.stft(y, 1024, 256, 1024)
    assert D.shape == (513, 1 + 5000 // 256) and D.dtype == np.complex64
    T = torch.stft(torch.from_numpy(y).double(), 1024, 256, 1024,
                   window=torch.hann_window(1024, periodic=True, dtype=torch.float64),
                   center=True, pad_mode='reflect', return_complex=True).numpy()
    np.testing.assert_allclose(D, T.astype(np.complex64), rtol=0, atol=2e-5)


def test_mel_silence_and_sine():
    m = mo.mel_spectrogram(np.zeros(4000, np.float32))
    assert m.shape == (1 + 4000 // 256, 80) and m.dtype == np.float32
    np.testing.assert_allclose(m, math.log(1e-5), rtol=1e-6)        # -11.512925
    t = np.arange(22050) / 22050.0
    y = (0.5 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    m = mo.mel_spectrogram(y)
    centre = mo.mel_frequencies(82, 0, 8000)[1:-1]
    assert abs(centre[int(m[40].argmax())] - 1000.0) < 60.0
    # fp32 pipeline agrees with an all-fp64 evaluation to ~1e-6 relative on linear mel
    me = mo.mel_spectrogram(y, exact=True)
    np.testing.assert_allclose(np.exp(m), np.exp(me), rtol=2e-5, atol=1e-7)


# ---------------------------------------------------------------- second, independent derivations of the mel path
def test_stft_matches_scipy_shorttimefft():
    """A second STFT implementation that shares nothing with the restatement (no frame gather, no np.fft call of ours):
    scipy.signal.ShortTimeFFT with the window start as phase origin, on the same reflect-padded signal."""
    import scipy.signal
    y = mo.synthetic_clip(7001, seed=3)
    D = mo.stft(y, 1024, 256, 1024)
    sft = scipy.signal.ShortTimeFFT(scipy.signal.get_window('hann', 1024, fftbins=True), hop=256, fs=1.0,
                                    fft_mode='onesided', mfft=1024, phase_shift=None)
    yp = np.pad(y.astype(np.float64), 512, mode='reflect')
    # slice p of ShortTimeFFT is centred on sample p * hop of its input = frame p - 2 of the padded signal
    S = sft.stft(yp, p0=2, p1=2 + D.shape[1])
    assert S.shape == D.shape
    np.testing.assert_allclose(D, S.astype(np.complex64), rtol=0, atol=3e-5)


# (mel filter, FFT bin, weight) of librosa.filters.mel(22050, 1024, n_mels=80, fmin=0, fmax=8000): derived a second time
# from the published definition with scalar arithmetic only - Slaney scale (200/3 Hz per mel below 1 kHz, log steps of
# ln(6.4)/27 above), 82 equally spaced mel points, triangle between neighbours, area normalisation 2 / (right - left) -
# and typed in; the zero entries pin the supports (which bins each filter may touch).
MEL_BASIS_SPOTS = [
    (0, 1, 1.5527720767e-02), (0, 2, 2.2651390211e-02), (1, 2, 4.2020256617e-03), (1, 3, 1.9729746429e-02),
    (10, 16, 0.0), (10, 17, 0.0), (39, 69, 0.0), (40, 72, 0.0), (40, 75, 0.0), (55, 140, 4.9497502311e-03),
    (60, 180, 0.0), (70, 262, 4.5833129691e-04), (79, 350, 1.4931705440e-03), (79, 360, 2.7828318605e-03),
    (79, 371, 1.2544655434e-04),
]


def test_mel_basis_spot_table_and_scalar_rederivation():
    B = mo.mel_filterbank(22050, 1024, 80, 0, 8000)
    for m, k, v in MEL_BASIS_SPOTS:
        assert abs(float(B[m, k]) - v) < 5e-9, (m, k, float(B[m, k]), v)
    # the whole matrix from the scalar form (no outer products, no vectorised ramps)
    h2m = lambda f: f / (200.0 / 3) if f < 1000.0 else 15.0 + math.log(f / 1000.0) * 27.0 / math.log(6.4)
    m2h = lambda m: m * (200.0 / 3) if m < 15.0 else 1000.0 * math.exp((m - 15.0) * math.log(6.4) / 27.0)
    hi = h2m(8000.0)
    pts = [m2h(hi * i / 81) for i in range(82)]
    worst = 0.0
    for m in range(80):
        l, c, r = pts[m], pts[m + 1], pts[m + 2]
        for k in range(513):
            f = k * 22050 / 1024
            w = 0.0 if (f <= l or f >= r) else ((f - l) / (c - l) if f <= c else (r - f) / (r - c)) * 2.0 / (r - l)
            worst = max(worst, abs(w - float(B[m, k])))
    assert worst < 5e-9
    # and the product's sparse form is the same matrix
    from transformertts_amd.data.audio import mel_filterbank_dense
    np.testing.assert_array_equal(mel_filterbank_dense(22050, 1024, 80, 0, 8000), B)


def test_group_assembly_equals_the_whole_batch():
    """tests/golden/make_config1_b32_golden.py builds the B = 32 golden from eight 4-sample fp64 runs (outputs
    concatenated, losses and gradients averaged, every group's expanded sequence zero-padded to the batch's
    `max_b sum(dur)`): the same assembly must equal ONE oracle run on the whole batch (reference
    model/models.py:464-482; the loss is an unmasked mean over the padded batch, utils/losses.py:41-49)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_config1_b32_golden as g32
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=5, perturb=0.02)
    batch = fo.synthetic_batch(8, 30, 120, seed=9, ragged=True)
    whole = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    whole.taps = []
    tr = whole.train_step(*batch, apply=False)
    parts = g32.run_groups(cfg, W, batch, group=2)
    assert parts['mel'].shape == tuple(tr['mel'].shape)
    np.testing.assert_allclose(parts['mel'], tr['mel'].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(parts['duration'], tr['duration'].numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(parts['loss'], float(tr['loss']), rtol=1e-12)
    np.testing.assert_allclose(parts['losses'], [float(tr['losses'][k]) for k in ('mel', 'duration', 'pitch')], rtol=1e-12)
    for (name, t) in whole.taps:
        np.testing.assert_allclose(parts['taps'][name], t.numpy(), rtol=0, atol=1e-12)
    for k, g in tr['grads'].items():
        np.testing.assert_allclose(parts['grads'][k], g.numpy(), rtol=0, atol=1e-12 + 1e-10 * float(g.abs().max()))


@pytest.mark.parametrize('n_fft,win,hop,fmin,fmax', [(1024, 1024, 256, 0.0, 8000.0), (2048, 1100, 275, 40.0, 11025.0)])
def test_mel_oracle_against_an_independent_restatement_of_librosa(n_fft, win, hop, fmin, fmax):
    """librosa is not installable here, but `transformers.audio_utils` (Hugging Face; in this image and on the GPU box) carries
    its own NumPy restatement of librosa's Slaney filterbank (`norm='slaney', mel_scale='slaney'`) and of its centred,
    reflect-padded STFT, written and tested against librosa by other people.  It agrees with oracle/mel_oracle.py: the
    filterbank to rounding, the log-mel of a clip to 1e-6 - for the LJSpeech / MelGAN setting and for the WaveRNN one
    (n_fft 2048, window 1100 zero-padded and centred, hop 275, fmin 40: config/data_config_wavernn.yaml:16-23).  A third
    restatement agreeing is evidence, not the reference's own vector: the row stays 'parity unpinned' in DESIGN.md."""
    au = pytest.importorskip('transformers.audio_utils')
    basis = mo.mel_filterbank(22050, n_fft, 80, fmin, fmax, dtype=np.float64)
    fb = au.mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=80, min_frequency=fmin, max_frequency=fmax,
                            sampling_rate=22050, norm='slaney', mel_scale='slaney')
    assert fb.shape == basis.T.shape
    assert np.abs(fb.T - basis).max() < 1e-12 * np.abs(basis).max() + 1e-15
    y = mo.synthetic_clip(20000 + hop, seed=3)
    w = au.window_function(win, 'hann', periodic=True, frame_length=n_fft, center=True)
    S = au.spectrogram(y.astype(np.float64), w, frame_length=n_fft, hop_length=hop, fft_length=n_fft, power=1.0, center=True,
                       pad_mode='reflect', mel_filters=fb, mel_floor=1e-5, log_mel='log', dtype=np.float64)
    exact = mo.mel_spectrogram(y, n_fft=n_fft, hop_length=hop, win_length=win, f_min=fmin, f_max=fmax, exact=True)
    assert S.T.shape == exact.shape
    assert np.abs(S.T - exact).max() < 1e-6
    fp32 = mo.mel_spectrogram(y, n_fft=n_fft, hop_length=hop, win_length=win, f_min=fmin, f_max=fmax)
    assert np.abs(S.T - fp32).max() < 2e-5
