#!/usr/bin/env python
"""Writes tests/golden/keras_mini_model_weights.hdf5 with the REAL libhdf5 (ctypes, tests/_libhdf5.py) in
the layout Keras `save_weights` gives the reference's ForwardTransformer (`model/models.py:600-618`).

TensorFlow / h5py are not installable here, so the file is not a TensorFlow product: the bytes are the
real HDF5 library's (same calls h5py makes: fixed-length `layer_names` / `weight_names` arrays,
variable-length `backend` / `keras_version`, UTF-8 attribute names, intermediate groups, contiguous
float32 datasets), the variable ORDER is a second, independent transcription of the reference
constructors (written out longhand below, not generated from
`transformertts_amd/model/keras_weights.py`), and the values are `oracle.ft_oracle.init_weights` of a
small config that has a dense block AND a conv block on each side.

    python tests/golden/make_keras_hdf5_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import _libhdf5 as H  # noqa: E402
from oracle import ft_oracle as fo  # noqa: E402

SEED = 4242


def mini_config():
    return fo.make_config(d_model=32, enc_heads=(1, 1), dec_heads=(1, 1), ffn=48, enc_dense_blocks=1,
                          dec_dense_blocks=1, conv_filters=(40, 32), conv_kernel=3, dur_filters=(24, 16),
                          pitch_filters=(24, 16))


def mini_weights():
    return fo.init_weights(mini_config(), seed=SEED, perturb=0.05)


def keras_order_longhand(W):
    """model.layers / layer.weights order of the reference, spelled out for THIS config (1 dense + 1 conv
    block per stack, two-layer predictors).  Names are placeholders; Keras loads by position."""
    def dense(scope, w, b):
        return [(f'{scope}/kernel:0', W[w]), (f'{scope}/bias:0', W[b])]

    def ln(scope, p):
        return [(f'{scope}/gamma:0', W[p + '.gamma']), (f'{scope}/beta:0', W[p + '.beta'])]

    def stack(name, r):
        s = f'forward_transformer/{name}'
        out = [('Variable:0', W[f'{r}.pos_scalar'])]                       # tf.Variable(1.) of the stack itself
        b = f'{s}/{name}_SADB_0'                                           # dense block 0: sarn then ffn
        out += dense(f'{b}/sarn/mha/wq', f'{r}.blk0.wq', f'{r}.blk0.bq')
        out += dense(f'{b}/sarn/mha/wk', f'{r}.blk0.wk', f'{r}.blk0.bk')
        out += dense(f'{b}/sarn/mha/wv', f'{r}.blk0.wv', f'{r}.blk0.bv')
        out += dense(f'{b}/sarn/mha/dense', f'{r}.blk0.wo', f'{r}.blk0.bo')
        out += ln(f'{b}/sarn/last_ln', f'{r}.blk0.ln1')
        out += dense(f'{b}/ffn/d1', f'{r}.blk0.ffn.w1', f'{r}.blk0.ffn.b1')
        out += dense(f'{b}/ffn/d2', f'{r}.blk0.ffn.w2', f'{r}.blk0.ffn.b2')
        out += ln(f'{b}/ffn/last_ln', f'{r}.blk0.ln2')
        b = f'{s}/{name}_SACB_0'                                           # conv block 0 (= blk1): sarn then conv
        out += dense(f'{b}/sarn/mha/wq', f'{r}.blk1.wq', f'{r}.blk1.bq')
        out += dense(f'{b}/sarn/mha/wk', f'{r}.blk1.wk', f'{r}.blk1.bk')
        out += dense(f'{b}/sarn/mha/wv', f'{r}.blk1.wv', f'{r}.blk1.bv')
        out += dense(f'{b}/sarn/mha/dense', f'{r}.blk1.wo', f'{r}.blk1.bo')
        out += ln(f'{b}/sarn/last_ln', f'{r}.blk1.ln1')
        out += [(f'{b}/conv/convolutions_0/kernel:0', W[f'{r}.blk1.conv0.w']), (f'{b}/conv/convolutions_0/bias:0', W[f'{r}.blk1.conv0.b'])]
        out += [(f'{b}/conv/last_conv/kernel:0', W[f'{r}.blk1.conv1.w']), (f'{b}/conv/last_conv/bias:0', W[f'{r}.blk1.conv1.b'])]
        out += ln(f'{b}/conv/normalization', f'{r}.blk1.ln2')
        out += ln(f'{s}/layernorm', f'{r}.ln')                             # the stack's input LayerNorm comes last
        return out

    def predictor(name, r):
        s = f'forward_transformer/{name}'
        out = [(f'{s}/conv_blocks/convolutions_0/kernel:0', W[f'{r}.conv0.w']), (f'{s}/conv_blocks/convolutions_0/bias:0', W[f'{r}.conv0.b'])]
        out += [(f'{s}/conv_blocks/last_conv/kernel:0', W[f'{r}.conv1.w']), (f'{s}/conv_blocks/last_conv/bias:0', W[f'{r}.conv1.b'])]
        out += ln(f'{s}/conv_blocks/normalization_0', f'{r}.ln0')
        out += ln(f'{s}/conv_blocks/normalization_1', f'{r}.ln1')
        out += dense(f'{s}/linear', f'{r}.lin.w', f'{r}.lin.b')
        return out

    return [
        ('Embedding', [('forward_transformer/Embedding/embeddings:0', W['embedding'])]),
        ('Encoder', stack('Encoder', 'enc')),
        ('dur_pred', predictor('dur_pred', 'dur')),
        ('expand', []),
        ('pitch_pred', predictor('pitch_pred', 'pitch')),
        ('dense_16', dense('forward_transformer/dense_16', 'pitch_embed.w', 'pitch_embed.b')),
        ('Decoder', stack('Decoder', 'dec')),
        ('dense_33', dense('forward_transformer/dense_33', 'out.w', 'out.b')),
    ]


def main():
    if H.find() is None:
        raise SystemExit('libhdf5 shared library not found (set TTSMI_LIBHDF5)')
    W = mini_weights()
    layers = keras_order_longhand(W)
    assert sum(len(ws) for _, ws in layers) == len(W)
    out = os.path.join(HERE, 'keras_mini_model_weights.hdf5')
    H.write_keras_weights(out, layers)
    print('wrote', out, os.path.getsize(out), 'bytes with libhdf5', H.version())


if __name__ == '__main__':
    main()
