"""CPU tests of the batch producer: tf.data bucket_by_sequence_length semantics as the
reference uses them (data/datasets.py:238-284), padding, determinism, endless reshuffled stream."""
import numpy as np
import pytest

from transformertts_amd.data.datasets import Dataset, TTSDataset, TTSPreprocessor, bucket_index, pad_and_stack


def _make(n=200, seed=0):
    rng = np.random.default_rng(seed)
    lens = rng.integers(50, 1300, size=n)
    store = {}
    for i, L in enumerate(lens):
        tp = max(2, int(L) // 5)
        dur = rng.multinomial(int(L), np.ones(tp) / tp).astype(np.int32)
        store[f's{i:04d}'] = (rng.standard_normal((int(L), 80)).astype(np.float32),
                              rng.integers(1, 100, tp).astype(np.int32), dur,
                              rng.standard_normal(tp).astype(np.float32), f's{i:04d}')
    return store


BOUNDS = [200, 300, 400, 500, 600, 700, 800, 900, 1000, 1200]        # config/training_config.yaml:22
SIZES = [64, 42, 32, 25, 21, 18, 16, 14, 12, 6, 1]                   # :23


def test_bucket_index_matches_tf_convention():
    assert bucket_index(10, BOUNDS) == 0 and bucket_index(199, BOUNDS) == 0
    assert bucket_index(200, BOUNDS) == 1 and bucket_index(299, BOUNDS) == 1
    assert bucket_index(1199, BOUNDS) == 9 and bucket_index(1200, BOUNDS) == 10 and bucket_index(5000, BOUNDS) == 10


def test_pad_and_stack():
    out = pad_and_stack([np.ones((3, 2), np.float32), np.ones((5, 2), np.float32)])
    assert out.shape == (2, 5, 2) and out[0, 3:].sum() == 0 and out.dtype == np.float32
    assert pad_and_stack(['a', 'b']) == ['a', 'b']
    assert pad_and_stack([np.int32(3), np.int32(4)]).tolist() == [3, 4]


@pytest.mark.parametrize('drop', [True, False])
def test_one_pass_bucket_semantics(drop):
    store = _make()
    ds = Dataset(list(store), store.__getitem__, lambda mel, *_: mel.shape[0], BOUNDS, SIZES, shuffle=True,
                 drop_remainder=drop, seed=7)
    seen = []
    for mel, tok, dur, pit, names in ds.all_batches():
        B = mel.shape[0]
        lens = [store[n][0].shape[0] for n in names]
        b = {bucket_index(L, BOUNDS) for L in lens}
        assert len(b) == 1                                            # a batch never mixes buckets
        bi = b.pop()
        assert B == SIZES[bi] or (not drop and B < SIZES[bi])
        assert mel.shape == (B, max(lens), 80) and mel.dtype == np.float32
        assert tok.dtype == np.int32 and dur.dtype == np.int32 and pit.dtype == np.float32
        assert tok.shape == dur.shape == pit.shape
        for i, n in enumerate(names):                                 # content and zero padding
            m0 = store[n][0]
            np.testing.assert_array_equal(mel[i, :m0.shape[0]], m0)
            assert not mel[i, m0.shape[0]:].any()
            assert dur[i].sum() == m0.shape[0]                        # sum(durations) == mel length survives padding
        seen += names
    assert len(seen) == len(set(seen))
    if not drop:
        assert sorted(seen) == sorted(store)                          # nothing lost without drop_remainder


def test_stream_is_deterministic_and_reshuffles_each_pass():
    store = _make(120, seed=3)
    mk = lambda: Dataset(list(store), store.__getitem__, lambda mel, *_: mel.shape[0], [400, 800], [8, 8, 4],  # noqa: E731
                         shuffle=True, drop_remainder=True, seed=11)
    a, b = mk(), mk()
    na = [a.next_batch()[4] for _ in range(40)]
    nb = [b.next_batch()[4] for _ in range(40)]
    assert na == nb                                                   # same seed, same stream
    per_pass = len(list(mk().all_batches()))
    assert na[:per_pass] != na[per_pass:2 * per_pass]                 # a new shuffle every pass
    a.close(); b.close()


def test_producer_errors_surface_in_next_batch():
    def bad(_):
        raise KeyError('missing sample')
    ds = Dataset(['x'], bad, lambda *a: 1, [10], [1, 1], seed=0)
    with pytest.raises(KeyError):
        ds.next_batch()
    ds.close()
    with pytest.raises(ValueError):
        Dataset([], bad, len, [10, 20], [1, 1])


def test_tts_dataset_reads_reference_layout(tmp_path):
    from transformertts_amd.data.text import Tokenizer
    tok = Tokenizer(add_start_end=False, model_breathing=False)
    store = _make(30, seed=5)
    for d in ('mels', 'durations', 'pitch'):
        (tmp_path / d).mkdir()
    text = {}
    alphabet = 'abdefhijklmnopstuvwz'
    for n, (mel, t, dur, pit, _) in store.items():
        np.save(tmp_path / 'mels' / f'{n}.npy', mel)
        np.save(tmp_path / 'durations' / f'{n}.npy', dur)
        np.save(tmp_path / 'pitch' / f'{n}.npy', pit)
        text[n] = ''.join(alphabet[int(x) % len(alphabet)] for x in t)
    ds = TTSDataset(text, TTSPreprocessor(80, tok), tmp_path / 'mels', tmp_path / 'durations', tmp_path / 'pitch')
    batches = list(ds.get_dataset([6, 6, 3], [400, 800], shuffle=False, drop_remainder=False).all_batches())
    assert sum(b[0].shape[0] for b in batches) == 30
    mel, phon, dur, pit, names = batches[0]
    assert phon.shape == dur.shape == pit.shape and mel.shape[2] == 80
    i = 0
    assert phon[i, :len(text[names[i]])].tolist() == tok(text[names[i]])


@pytest.mark.gpu
def test_device_staging_feeds_a_train_step():
    """Producer thread -> pinned buffers -> async copy on its own stream -> train_step on the GPU."""
    import torch
    from oracle import ft_oracle as fo
    from transformertts_amd.model.models import ForwardTransformer
    rng = np.random.default_rng(0)
    store = {}
    for i in range(24):
        tp = int(rng.integers(10, 40))
        dur = rng.integers(1, 5, tp).astype(np.int32)
        store[f'u{i}'] = (rng.standard_normal((int(dur.sum()), 80)).astype(np.float32),
                          rng.integers(1, 100, tp).astype(np.int32), dur,
                          rng.standard_normal(tp).astype(np.float32), f'u{i}')
    ds = Dataset(list(store), store.__getitem__, lambda mel, *_: mel.shape[0], [60], [4, 4], seed=1,
                 device='cuda:0', prefetch=2)
    model = ForwardTransformer.from_config(dict(fo.tiny_config(), device='cuda:0', seed=0))
    model._compile(learning_rate=1e-3)
    for _ in range(5):
        mel, phon, dur, pit, names = ds.next_batch()
        assert mel.is_cuda and phon.is_cuda and mel.dtype == torch.float32 and phon.dtype == torch.int32
        i = 0
        np.testing.assert_array_equal(mel[i, :store[names[i]][0].shape[0]].cpu().numpy(), store[names[i]][0])
        out = model.train_step(phon, mel, dur, pit)
        assert np.isfinite(float(out['loss']))
    ds.close()
