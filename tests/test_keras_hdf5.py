"""Keras `model_weights.hdf5` interchange: the dependency-free HDF5 reader /
writer against the REAL libhdf5 (committed fixture written by it; live cross-checks through ctypes when
the shared library is present), and the Keras variable-order table against an independent longhand
transcription and the oracle's weight spec."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, 'golden'))

import _libhdf5 as H  # noqa: E402
import make_keras_hdf5_fixture as mk  # noqa: E402
from oracle import ft_oracle as fo  # noqa: E402
from transformertts_amd.model.keras_weights import (HDF5_OBJECT_HEADER_LIMIT, keras_layer_table,  # noqa: E402
                                                    load_keras_weights, save_keras_weights)
from transformertts_amd.utils import hdf5_min as M  # noqa: E402

FIXTURE = os.path.join(HERE, 'golden', 'keras_mini_model_weights.hdf5')
needs_libhdf5 = pytest.mark.skipif(H.find() is None, reason='libhdf5 shared library not found')


def _f32(W):
    return {k: np.asarray(v, dtype=np.float32) for k, v in W.items()}


# ------------------------------------------------------------------------------ reader vs the real library
def test_reader_on_fixture_written_by_real_libhdf5():
    W = _f32(mk.mini_weights())
    with M.File(FIXTURE) as f:
        assert f.attrs['backend'] == 'tensorflow' and f.attrs['keras_version'] == '2.4.0'      # variable-length UTF-8
        names = [n.decode() for n in f.attrs['layer_names']]                                 # fixed-length array
        assert names == ['Embedding', 'Encoder', 'dur_pred', 'expand', 'pitch_pred', 'dense_16', 'Decoder', 'dense_33']
        assert len(f['expand'].attrs['weight_names']) == 0
        long = mk.keras_order_longhand(W)
        for lname, ws in long:
            g = f[lname]
            assert [n.decode() for n in g.attrs['weight_names']] == [n for n, _ in ws]
            for n, a in ws:
                d = g[n]
                assert d.shape == np.shape(a) and d.dtype == np.dtype('<f4')
                np.testing.assert_array_equal(np.asarray(d), a)
        assert f['Encoder/Variable:0'].shape == () and float(f['Encoder/Variable:0'][()]) == float(W['enc.pos_scalar'])
        assert 'Encoder/forward_transformer/Encoder/layernorm' in f and 'Encoder/nope' not in f
        with pytest.raises(KeyError):
            f['Encoder/nope']


def test_fixture_loads_into_package_names_bit_exact():
    cfg, W = mk.mini_config(), _f32(mk.mini_weights())
    got = load_keras_weights(FIXTURE, cfg, fo.VOCAB_SIZE)
    assert set(got) == set(W)
    for k in W:
        assert got[k].dtype == np.float32 and got[k].shape == W[k].shape
        np.testing.assert_array_equal(got[k], W[k])


@needs_libhdf5
@pytest.mark.parametrize('kw', [{}, {'chunked': True}, {'libver_latest': True}])
def test_reader_on_files_written_now_by_real_libhdf5(tmp_path, kw):
    """Default h5py settings (v0 superblock, v1 headers, link messages because of the UTF-8 link names),
    chunked datasets (v1 B-tree chunk index) and libver='latest' (v2 headers, v4 layouts, v3 superblock)."""
    rng = np.random.default_rng(3)
    layers = [('L0', [('m/L0/kernel:0', rng.standard_normal((5, 7)).astype('f4')), ('m/L0/bias:0', rng.standard_normal(7).astype('f4'))]),
              ('empty', []),
              ('L1', [('Variable:0', np.float32(0.25)), ('m/L1/conv/kernel:0', rng.standard_normal((3, 6, 10)).astype('f4'))]),
              ('many', [(f'm/many/w{i}:0', rng.standard_normal((i % 3 + 1, 2)).astype('f4')) for i in range(7)])]
    p = str(tmp_path / 'w.h5')
    H.write_keras_weights(p, layers, **kw)
    with M.File(p) as f:
        assert [n.decode() for n in f.attrs['layer_names']] == [n for n, _ in layers]
        for lname, ws in layers:
            g = f[lname]
            assert [n.decode() for n in g.attrs['weight_names']] == [n for n, _ in ws]
            for n, a in ws:
                np.testing.assert_array_equal(np.asarray(g[n]), a)
                assert g[n].shape == np.shape(a)


@needs_libhdf5
@pytest.mark.parametrize('utf8_links', [True, False])
def test_reader_walks_large_groups(tmp_path, utf8_links):
    """70 links in one group.  UTF-8 link names (h5py's default): link messages in a fractal heap that
    spans several direct blocks under an indirect root block.  ASCII link names (plain C API): an
    old-style group whose v1 B-tree points at several symbol nodes (2K = 8 entries each)."""
    rng = np.random.default_rng(5)
    ws = [(f'v{i:03d}:0', rng.standard_normal(3).astype('f4')) for i in range(70)]
    p = str(tmp_path / 'big.h5')
    H.write_keras_weights(p, [('big', ws)], utf8_links=utf8_links)
    with M.File(p) as f:
        kinds = {m.type for m in f['big']._msgs}
        assert (0x02 in kinds) == utf8_links and (0x11 in kinds) == (not utf8_links)
        assert f['big'].keys() == sorted(n for n, _ in ws)
        for n, a in ws:
            np.testing.assert_array_equal(np.asarray(f['big'][n]), a)


@needs_libhdf5
def test_ten_block_stack_from_real_libhdf5_dense_link_storage(tmp_path):
    """h5py names links in UTF-8, so libhdf5 keeps them as link messages - and moves them to a fractal
    heap once a group has more than 8: a stack of 10 blocks has 11 sub-groups under
    /Encoder/forward_transformer/Encoder.  Written by the C library, loaded through the order table."""
    cfg = fo.make_config(d_model=32, enc_heads=(1,) * 10, dec_heads=(1,) * 10, ffn=40, dur_filters=(16, 16),
                         pitch_filters=(16, 16))
    W = _f32(fo.init_weights(cfg, seed=9, perturb=0.1))
    layers = [(ln, [(k, W[r]) for k, r, _ in es]) for ln, es in keras_layer_table(cfg, fo.VOCAB_SIZE)]
    p = str(tmp_path / 'ten.hdf5')
    H.write_keras_weights(p, layers)
    with M.File(p) as f:
        assert len(f['Encoder/forward_transformer/Encoder'].keys()) == 11
    got = load_keras_weights(p, cfg, fo.VOCAB_SIZE)
    for k in W:
        np.testing.assert_array_equal(got[k], W[k])


# ------------------------------------------------------------------------------ writer vs the real library
@needs_libhdf5
def test_real_libhdf5_reads_what_the_writer_wrote(tmp_path):
    cfg, W = mk.mini_config(), _f32(mk.mini_weights())
    p = str(tmp_path / 'model_weights.hdf5')
    save_keras_weights(p, W, cfg, fo.VOCAB_SIZE)
    got, meta = H.read_keras_weights(p)                       # H5Fopen / H5Aread / H5Dread of the C library
    assert meta == {'backend': b'tensorflow', 'keras_version': b'2.4.0'}
    long = mk.keras_order_longhand(W)
    assert [n for n, _ in got][:5] == [n for n, _ in long][:5]
    for (lname, ws), (_, ws2) in zip(got, long):
        assert len(ws) == len(ws2), lname
        for (n, a), (_, b) in zip(ws, ws2):
            assert a.shape == np.shape(b), n
            np.testing.assert_array_equal(a, b)


def test_writer_reader_roundtrip_and_attribute_chunking(tmp_path):
    """Keras splits name lists that do not fit a 64 KiB object header into name0, name1, ...; groups with
    hundreds of entries; scalar, empty and integer data."""
    w = M.Writer()
    g = w.root.create_group('g')
    names = [(f'some/deeply/nested/scope_{i:04d}/' + 'x' * 60 + '/kernel:0').encode() for i in range(900)]
    assert sum(len(n) for n in names) > HDF5_OBJECT_HEADER_LIMIT
    from transformertts_amd.model import keras_weights as kw
    kw._save_attribute(g, 'weight_names', names)
    assert 'weight_names' not in g.attrs and 'weight_names0' in g.attrs and 'weight_names1' in g.attrs
    for i in range(300):
        g.create_dataset(f'flat{i:03d}', np.full((2,), i, np.float32))
    g.create_dataset('a/b/c', np.arange(24, dtype=np.float32).reshape(2, 3, 4))
    g.create_dataset('scalar', np.float32(3.5))
    g.create_dataset('empty', np.zeros((0, 4), np.float32))
    g.create_dataset('ints', np.arange(5, dtype=np.int64))
    w.root.attrs['note'] = 'plain ascii'
    p = str(tmp_path / 'rt.h5')
    w.save(p)
    with M.File(p) as f:
        assert kw._load_attribute(f['g'], 'weight_names') == [n.decode() for n in names]
        assert f.attrs['note'] == b'plain ascii'
        assert len(f['g'].keys()) == 300 + 4
        np.testing.assert_array_equal(np.asarray(f['g/flat123']), [123, 123])
        np.testing.assert_array_equal(np.asarray(f['g/a/b/c']), np.arange(24, dtype=np.float32).reshape(2, 3, 4))
        assert f['g/scalar'].shape == () and float(np.asarray(f['g/scalar'])) == 3.5
        assert np.asarray(f['g/empty']).shape == (0, 4)
        assert np.asarray(f['g/ints']).dtype == np.int64 and np.asarray(f['g/ints']).tolist() == [0, 1, 2, 3, 4]
    if H.find() is not None:                                   # and the C library agrees on the big group
        L = H.find()
        fid = L.H5Fopen(p.encode(), 0, 0)
        gid = L.H5Gopen2(fid, b'g', 0)
        np.testing.assert_array_equal(H._read_dataset_f32(gid, 'flat299'), [299, 299])
        np.testing.assert_array_equal(H._read_dataset_f32(gid, 'a/b/c').ravel(), np.arange(24))
        assert H._read_attr_strings(gid, 'weight_names1')[0] in names
        L.H5Gclose(gid), L.H5Fclose(fid)


# ------------------------------------------------------------------------------ the Keras order table
@pytest.mark.parametrize('cfg', [fo.make_config(), fo.tiny_config(), mk.mini_config(),
                                 fo.make_config(d_model=384, enc_heads=(2,) * 6, dec_heads=(2,) * 6,
                                                enc_dense_blocks=0, dec_dense_blocks=0, conv_filters=(1536, 384))],
                         ids=['bench', 'tiny', 'mini', 'reference-default-conv'])
def test_order_table_covers_the_weight_spec(cfg):
    table = keras_layer_table(cfg, fo.VOCAB_SIZE)
    spec = fo.weight_spec(cfg)
    flat = [e for _, es in table for e in es]
    assert sorted(r for _, r, _ in flat) == sorted(spec)                         # every variable exactly once
    for kname, ref, shape in flat:
        assert tuple(spec[ref]) == tuple(shape), ref
    assert [n for n, _ in table][:5] == ['Embedding', 'Encoder', 'dur_pred', 'expand', 'pitch_pred']
    assert [n for n, _ in table][6] == 'Decoder' and table[3][1] == []
    enc = dict(table)['Encoder']
    assert enc[0][:2] == ('Variable:0', 'enc.pos_scalar')                        # the stack's own tf.Variable first
    assert [r for _, r, _ in enc[-2:]] == ['enc.ln.gamma', 'enc.ln.beta']       # its LayerNorm last
    assert [r.split('.')[-1] for _, r, _ in enc[1:9]] == ['wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'wo', 'bo']
    dur = [r for _, r, _ in dict(table)['dur_pred']]
    assert dur[:4] == ['dur.conv0.w', 'dur.conv0.b', 'dur.conv1.w', 'dur.conv1.b'] and dur[-2:] == ['dur.lin.w', 'dur.lin.b']
    dsets = [f'{ln}/{k}' for ln, es in table for k, _, _ in es]
    assert len(set(dsets)) == len(dsets)                                         # no two variables share a dataset path
    # Keras' auto-naming: the two top-level Dense layers are numbered after every Dense built before them
    n_dense = lambda stack: sum(1 for k, _, _ in dict(table)[stack] if k.endswith('kernel:0') and '/dense' in k)
    assert table[5][0] == f'dense_{n_dense("Encoder") + 2}'


def test_order_table_matches_longhand_transcription():
    W = _f32(mk.mini_weights())
    table = keras_layer_table(mk.mini_config(), fo.VOCAB_SIZE)
    long = mk.keras_order_longhand(W)
    for (ln, es), (ln2, ws) in zip(table, long):
        assert len(es) == len(ws), ln
        for (_, ref, shape), (_, a) in zip(es, ws):
            assert a is W[ref] or np.array_equal(a, W[ref]), (ln, ref)


def test_load_errors_are_loud(tmp_path):
    cfg = mk.mini_config()
    with pytest.raises(ValueError, match='expects .* weights|has shape'):
        load_keras_weights(FIXTURE, fo.make_config(d_model=32, enc_heads=(1, 1), dec_heads=(1, 1), ffn=64,
                                                   dur_filters=(24, 16), pitch_filters=(24, 16)), fo.VOCAB_SIZE)
    with pytest.raises(ValueError, match='has shape'):
        load_keras_weights(FIXTURE, cfg, fo.VOCAB_SIZE + 1)
    raw = open(FIXTURE, 'rb').read()
    p = tmp_path / 'trunc.hdf5'
    p.write_bytes(raw[:len(raw) // 2])
    with pytest.raises(M.Hdf5Error, match='outside the file'):
        load_keras_weights(str(p), cfg, fo.VOCAB_SIZE)
    p = tmp_path / 'junk.hdf5'
    p.write_bytes(b'not an hdf5 file' * 100)
    with pytest.raises(M.Hdf5Error, match='not an HDF5 file'):
        M.File(str(p))
    W = _f32(mk.mini_weights())
    W['enc.blk0.wq'] = W['enc.blk0.wq'][:, :8]
    with pytest.raises(ValueError, match='enc.blk0.wq'):
        save_keras_weights(str(tmp_path / 'bad.hdf5'), W, cfg, fo.VOCAB_SIZE)


# ------------------------------------------------------------------------------ property test of the container format
def test_random_trees_roundtrip_and_are_readable_by_real_libhdf5(tmp_path):
    """Seeded random group trees (nested groups, names with ':' / spaces / non-ASCII, float and integer datasets of
    rank 0-3, string and numeric attributes): writer -> reader is the identity, and the C library opens every
    float dataset of the same files."""
    rng = np.random.default_rng(2024)
    alphabet = list('abcXYZ_:0189 -') + ['é', 'ß', 'ʃ']
    L = H.find()
    for trial in range(12):
        w = M.Writer()
        expect = {}

        def name():
            return ''.join(rng.choice(alphabet, size=int(rng.integers(1, 9)))).strip() or 'n'

        def fill(g, path, depth):
            for _ in range(int(rng.integers(0, 6 if depth else 9))):
                nm = name()
                if nm in g.children or '/' in nm or nm in ('.', '..'):
                    continue
                if depth < 3 and rng.random() < 0.35:
                    sub = g.create_group(nm)
                    sub.attrs['note'] = np.array([name().encode('utf8') for _ in range(int(rng.integers(1, 4)))])
                    fill(sub, path + '/' + nm, depth + 1)
                else:
                    shape = tuple(int(x) for x in rng.integers(0, 5, size=int(rng.integers(0, 4))))
                    dt = [np.float32, np.float64, np.int32, np.int64][int(rng.integers(0, 4))]
                    a = (rng.standard_normal(shape) * 100).astype(dt)
                    ds = g.create_dataset(nm, a)
                    ds.attrs['scale'] = np.float64(rng.standard_normal())
                    expect[path + '/' + nm] = a
        fill(w.root, '', 0)
        w.root.attrs['trial'] = np.int64(trial)
        p = str(tmp_path / f't{trial}.h5')
        w.save(p)
        with M.File(p) as f:
            assert int(f.attrs['trial']) == trial
            for path, a in expect.items():
                d = f[path]
                got = np.asarray(d)
                assert got.dtype == a.dtype and got.shape == a.shape, path
                np.testing.assert_array_equal(got, a)
                assert isinstance(float(d.attrs['scale']), float)
        if L is not None:
            fid = L.H5Fopen(p.encode(), 0, 0)
            assert fid >= 0
            for path, a in expect.items():
                if a.dtype == np.float32 and a.size:
                    np.testing.assert_array_equal(H._read_dataset_f32(fid, path.lstrip('/')), a)
            L.H5Fclose(fid)
