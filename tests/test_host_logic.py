"""CPU tests of the host-side logic (no kernels run): constants handed to the kernels equal the
oracle's, parameter layout, tokenizer, loud failure without a GPU."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo
from oracle import mel_oracle as mo


def test_sparse_filterbank_equals_oracle_dense_basis():
    from transformertts_amd.data.audio import hann_window_padded, mel_filterbank_sparse
    for (sr, n_fft, n_mels, fmin, fmax) in [(22050, 1024, 80, 0, 8000), (22050, 2048, 80, 40, None),
                                             (16000, 1024, 40, 0, None)]:
        lo, cnt, ptr, w = mel_filterbank_sparse(sr, n_fft, n_mels, fmin, fmax)
        dense = np.zeros((n_mels, 1 + n_fft // 2), np.float32)
        for m in range(n_mels):
            dense[m, lo[m]:lo[m] + cnt[m]] = w[ptr[m]:ptr[m] + cnt[m]]
        np.testing.assert_array_equal(dense, mo.mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    assert len(mel_filterbank_sparse(22050, 1024, 80, 0, 8000)[3]) >= 727
    import scipy.signal
    np.testing.assert_allclose(hann_window_padded(1024, 1024),
                               scipy.signal.get_window('hann', 1024, fftbins=True), atol=1e-7)
    w = hann_window_padded(1100, 2048)
    assert w[:474].sum() == 0 and w[474 + 1100:].sum() == 0
    np.testing.assert_allclose(w[474:474 + 1100], scipy.signal.get_window('hann', 1100, fftbins=True), atol=1e-7)


def test_wav_preprocessing_steps_that_fix_the_frame_count():
    """Reference Audio.preprocess (data/audio.py:132-141): volume normalisation (increase only) and the
    one-sample pad when len % hop == 0."""
    from transformertts_amd.data.audio import normalize_volume, pad_for_frame_count
    y = np.arange(512, dtype=np.float32)
    assert pad_for_frame_count(y, 256).shape == (513,) and pad_for_frame_count(y, 256)[-1] == 0
    assert pad_for_frame_count(y[:500], 256) is not None and pad_for_frame_count(y[:500], 256).shape == (500,)
    assert 1 + len(pad_for_frame_count(y, 256)) // 256 == 3
    rng = np.random.default_rng(0)
    w = (0.01 * rng.standard_normal(4000)).astype(np.float32)
    out = normalize_volume(w, target_dBFS=-30, int16_max=32767, increase_only=True)
    rms_db = 20 * np.log10(np.sqrt(np.mean((out * 32767) ** 2)) / 32767)
    assert abs(rms_db + 30) < 1e-4                                 # -40 dBFS clip raised to the target
    loud = (0.5 * rng.standard_normal(4000)).astype(np.float32)
    assert normalize_volume(loud, -30, 32767, increase_only=True) is loud      # never attenuated
    assert normalize_volume(loud, -30, 32767, decrease_only=True) is not loud
    with pytest.raises(ValueError):
        normalize_volume(w, -30, 32767, increase_only=True, decrease_only=True)


def test_flat_params_layout_and_spec_matches_oracle():
    from transformertts_amd.model.models import FlatParams, _blocks_spec, _predictor_spec
    spec = OrderedDict()
    spec['a'] = (3, 5)
    spec['s'] = ()
    spec['b'] = (7,)
    P = FlatParams(spec, 'cpu')
    assert P.offsets['a'] == (0, 15) and P.offsets['s'] == (16, 1) and P.offsets['b'] == (24, 7)
    assert P.total == 32 and P.n_params == 23          # every tensor starts on a 32-byte boundary
    P.g['b'].fill_(2.0)
    assert P.grad[24:31].eq(2).all() and P.grad[:24].eq(0).all()
    assert P.w['a'].requires_grad and P.w['a'].data_ptr() == P.data.data_ptr()
    # parameter count of the benchmark config equals the oracle's spec (SURVEY: 11.06 M)
    cfg = fo.make_config()
    n_or = sum(int(np.prod(s)) for s in fo.weight_spec(cfg).values())
    mine = OrderedDict()
    mine['embedding'] = (127, 256)
    mine.update(_blocks_spec('enc', 256, [4] * 6, 6, 1024, None, 3))
    mine.update(_predictor_spec('dur', 256, [256, 226], 3))
    mine.update(_predictor_spec('pitch', 256, [256, 226], 3))
    mine['pitch_embed.w'] = (1, 256)
    mine['pitch_embed.b'] = (256,)
    mine.update(_blocks_spec('dec', 256, [4] * 6, 6, 1024, None, 3))
    mine['out.w'] = (256, 80)
    mine['out.b'] = (80,)
    n_mine = sum(int(np.prod(s)) for s in mine.values())
    assert n_mine == n_or
    assert 11.0e6 < n_or < 11.1e6


def test_tokenizer_matches_reference_semantics():
    from transformertts_amd.data.text import Tokenizer
    t = Tokenizer(add_start_end=False, model_breathing=False)
    assert t.vocab_size == 127 and t.decode([0]) == '/'
    assert t.decode(t('hɛloʊ')) == 'hɛloʊ'
    # reference tests/test_char_tokenizer.py alphabet 'ab c' with start/end + breathing defaults
    t2 = Tokenizer(alphabet=list('ab c'))
    assert t2.start_token_index == 5 and t2.end_token_index == 6 and t2.vocab_size == 8
    with pytest.raises(KeyError):
        t2('a b d')
    t3 = Tokenizer(alphabet=list('ab c'), model_breathing=False)
    assert t3.vocab_size == 7 and t3('a b') == [5, 2, 1, 3, 6]


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from transformertts_amd import _lib
    from transformertts_amd.data.audio import Audio
    from transformertts_amd.model.models import ForwardTransformer
    with pytest.raises(_lib.TtsmiError):
        ForwardTransformer.from_config(fo.tiny_config())
    with pytest.raises(_lib.TtsmiError):
        Audio(22050, 1024, 80, 256, 1024, 0, 8000, 'MelGAN')
    from transformertts_amd import ops
    with pytest.raises(_lib.TtsmiError):
        ops.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4), None)


def test_product_code_never_imports_the_oracle():
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'transformertts_amd')
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), os.path.join(dp, f)


def test_phonemizer_adapter_uses_the_dependency_when_present_and_raises_without_it(monkeypatch):
    """The reference's text entry point (data/text/tokenizer.py:50-104, data/text/__init__.py:18-21): with
    `phonemizer` importable the adapter calls it with the reference's arguments and cleans the result; without it (this
    image) a clear error, not an ImportError at module import."""
    import sys
    import types
    from transformertts_amd.data.text import Phonemizer, TextToTokens
    for name in ('phonemizer', 'phonemizer.phonemize'):
        monkeypatch.delitem(sys.modules, name, raising=False)
    tt = TextToTokens.default('en', True, True, True)
    assert isinstance(tt.phonemizer, Phonemizer)
    try:
        import phonemizer  # noqa: F401
        have = True
    except ImportError:
        have = False
    if not have:
        with pytest.raises(RuntimeError, match='phonemizer'):
            tt('hello')
    seen = {}

    def fake(texts, **kw):
        seen.update(kw, texts=list(texts))
        return ['h ə l oʊ  ,  w ɜː l d — X 9 !'] * len(texts)
    pkg, mod = types.ModuleType('phonemizer'), types.ModuleType('phonemizer.phonemize')
    mod.phonemize = fake
    monkeypatch.setitem(sys.modules, 'phonemizer', pkg)
    monkeypatch.setitem(sys.modules, 'phonemizer.phonemize', mod)
    ph = Phonemizer('en-us', with_stress=True, njobs=2)
    assert ph('well-known, world!') == 'h ə l oʊ,w ɜː l d-!'          # unknown symbols dropped, spaces around marks gone
    assert seen['texts'] == ['well—known, world!'] and seen['backend'] == 'espeak' and seen['language'] == 'en-us'
    assert seen['with_stress'] is True and seen['njobs'] == 2 and seen['preserve_punctuation'] and seen['strip']
    assert ph(['a', 'b']) == ['h ə l oʊ,w ɜː l d-!'] * 2
    ids = tt('hello')
    assert ids[0] == tt.tokenizer.start_token_index and ids[-1] == tt.tokenizer.end_token_index
    with pytest.raises(TypeError):
        ph(3)


def test_autograd_keeps_the_first_gradient_by_reference_and_runs_the_producer_last():
    """What ops.GradSink relies on, checked against the installed torch with three CPU Functions of the same shape as the
    per-layer path: a tensor h feeds a consumer and, as the residual, the node that consumes the consumer's output.  The
    residual node's backward runs first, returns its gradient for h and keeps the tensor; the consumer's backward adds its own
    contribution INTO that tensor and returns None.  The producer of h must then be handed the sum - i.e. autograd stored the
    first gradient by reference (no copy) and called the producer's backward only after every consumer's."""
    import torch
    sink, seen, order = {}, [], []

    class Residual(torch.autograd.Function):              # like AddLayerNormFn with res_sink
        @staticmethod
        def forward(ctx, x, res):
            return x + res

        @staticmethod
        def backward(ctx, g):
            order.append('residual')
            dres = g.clone()
            sink['buf'] = dres
            return g, dres

    class Consumer(torch.autograd.Function):              # like LinearFn with x_sink: y = 2 h
        @staticmethod
        def forward(ctx, h):
            return 2 * h

        @staticmethod
        def backward(ctx, g):
            order.append('consumer')
            sink['buf'].add_(2 * g)
            return None

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            order.append('producer')
            seen.append(g.clone())
            return g

    x = torch.ones(5, requires_grad=True)
    h = Producer.apply(x)
    y = Residual.apply(Consumer.apply(h), h)
    y.sum().backward()
    assert order == ['residual', 'consumer', 'producer']
    assert torch.equal(seen[0], torch.full((5,), 3.0)) and torch.equal(x.grad, torch.full((5,), 3.0))
