"""Keras `model_weights.hdf5` interchange for ForwardTransformer.

The reference saves with `self.save_weights(path / 'model_weights.hdf5')` and restores with
`model.load_weights(...)` (`model/models.py:600-638`).  For an `.hdf5` path Keras uses its "legacy" H5
layout (`tensorflow/python/keras/saving/hdf5_format.py`, `save_weights_to_hdf5_group` /
`load_weights_from_hdf5_group`):

  /                       attrs  layer_names   = [layer.name for layer in model.layers]
                                 backend, keras_version
  /<layer.name>           attrs  weight_names  = [w.name for w in layer.trainable_weights + layer.non_trainable_weights]
  /<layer.name>/<w.name>  one float32 dataset per variable (variable names contain '/', so they nest)

and **loads by position, not by name**: the layers that own weights are zipped with the groups that
hold weights, and inside a group the datasets are taken in `weight_names` order and assigned to the
layer's weights in `layer.weights` order; only counts and shapes are checked.  What has to be
reproduced is therefore the ORDER Keras enumerates the reference's variables in, which follows from
the attribute order of its constructors:

  * `model.layers` = the layers assigned in `ForwardTransformer.__init__` (`model/models.py:377-421`):
    encoder_prenet 'Embedding', encoder 'Encoder', dur_pred, expand (no weights), pitch_pred,
    pitch_embed (a Dense), decoder 'Decoder', out (a Dense);
  * a layer's weights = its own variables, then its sub-layers' in attribute order, recursively.
    `SelfAttentionBlocks` (`model/layers.py:267-295`) owns `pos_encoding_scalar` (a bare
    `tf.Variable`, so it comes FIRST), then the dense blocks, the conv blocks, and its input
    LayerNormalization LAST.  A block is sarn (mha: wq, wk, wv, dense; then last_ln) followed by ffn
    (d1, d2, last_ln; `layers.py:82-96`) or conv (convolutions..., last_conv, normalization;
    `layers.py:6-28`).  `StatPredictor` (`layers.py:463-508`) is conv_blocks (convolutions...,
    last_conv, then ALL LayerNormalizations) followed by linear.  Dense / Conv1D contribute
    (kernel, bias), LayerNormalization (gamma, beta), Embedding (embeddings).

The table below spells that order out against this package's variable names (the ones
`ForwardTransformer.weights_dict()` emits and `oracle/ft_oracle.py:weight_spec` lists).  TensorFlow is
not available offline, so no file written by TensorFlow pins the order; what does check it: (i) the table is
used to load weights into the reference's OWN `ForwardTransformer`, constructed from the reference source over a
stand-in for TensorFlow whose layer base class implements Keras' tracking rule (own variables first, then
sub-layers in attribute-assignment order) - layer count, per-layer weight count and every slot's shape must
match the layers the reference constructors really create, and the loaded model then reproduces the oracle's
outputs and gradients to 1e-10 (tests/golden/make_reference_source_run.py, tests/test_reference_source_run.py);
(ii) an independent longhand transcription of the order agrees (tests/golden/make_keras_hdf5_fixture.py);
(iii) every slot is shape-checked on load.  Variable names written on
save follow TensorFlow's naming scheme (`forward_transformer/Encoder/Encoder_SADB_0/.../dense/kernel:0`,
Keras' `unique_object_name` counters in construction order); Keras never reads them back.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np

from ..utils import hdf5_min

# Keras splits attributes that would not fit an HDF5 object header (hdf5_format.py HDF5_OBJECT_HEADER_LIMIT)
HDF5_OBJECT_HEADER_LIMIT = 64512

Entry = Tuple[str, str, tuple]          # (keras variable name, this package's variable name, shape)


class _Uid:
    """Keras `backend.unique_object_name(..., zero_based=True)`: 'dense', 'dense_1', 'dense_2', ..."""

    def __init__(self):
        self.count: Dict[str, int] = {}

    def __call__(self, base: str) -> str:
        n = self.count.get(base, 0)
        self.count[base] = n + 1
        return base if n == 0 else f'{base}_{n}'


def _dense(uid, scope, ref_w, ref_b, cin, cout) -> List[Entry]:
    n = uid('dense')
    return [(f'{scope}/{n}/kernel:0', ref_w, (cin, cout)), (f'{scope}/{n}/bias:0', ref_b, (cout,))]


def _layernorm(uid, scope, ref, c) -> List[Entry]:
    n = uid('layer_normalization')
    return [(f'{scope}/{n}/gamma:0', ref + '.gamma', (c,)), (f'{scope}/{n}/beta:0', ref + '.beta', (c,))]


def _blocks(uid, model_scope, name, ref, d, heads, dense_blocks, ffn, conv_filters, conv_kernel, transposed) -> List[Entry]:
    """`SelfAttentionBlocks(name=name)` in Keras weight order; construction order drives the name counters."""
    scope = f'{model_scope}/{name}'
    entries: List[Entry] = [('Variable:0', f'{ref}.pos_scalar', ())]      # created in __init__, outside any scope
    uid('dropout')
    for i in range(len(heads)):
        dense_block = i < dense_blocks
        p = f'{ref}.blk{i}'
        bscope = f'{scope}/{name}_{"SADB" if dense_block else "SACB"}_{i if dense_block else i - dense_blocks}'
        sarn = f'{bscope}/{uid("self_attention_res_norm")}'
        mha = f'{sarn}/{uid("multi_head_attention")}'
        entries += _dense(uid, mha, f'{p}.wq', f'{p}.bq', d, d)
        entries += _dense(uid, mha, f'{p}.wk', f'{p}.bk', d, d)
        entries += _dense(uid, mha, f'{p}.wv', f'{p}.bv', d, d)
        uid('scaled_dot_product_attention'), uid('dropout')
        entries += _dense(uid, mha, f'{p}.wo', f'{p}.bo', 2 * d, d)      # Dense(concat([q_in, ctx])) layers.py:148-149
        uid('dropout')
        entries += _layernorm(uid, sarn, f'{p}.ln1', d)
        if dense_block:
            fscope = f'{bscope}/{uid("ffn_res_norm")}'
            entries += _dense(uid, fscope, f'{p}.ffn.w1', f'{p}.ffn.b1', d, ffn)
            entries += _dense(uid, fscope, f'{p}.ffn.w2', f'{p}.ffn.b2', ffn, d)
            uid('dropout')
            entries += _layernorm(uid, fscope, f'{p}.ln2', d)
        else:
            cscope = f'{bscope}/{uid("transposed_cnn_res_norm" if transposed else "cnn_res_norm")}'
            cin = d
            convs = []
            for j, f in enumerate(conv_filters[:-1]):
                convs.append((uid('conv1d'), j, cin, f))
                cin = f
            for _ in conv_filters[:-1]:
                uid('activation')
            convs.append((uid('conv1d'), len(conv_filters) - 1, cin, conv_filters[-1]))
            for n, j, ci, co in convs:
                entries += [(f'{cscope}/{n}/kernel:0', f'{p}.conv{j}.w', (conv_kernel, ci, co)),
                            (f'{cscope}/{n}/bias:0', f'{p}.conv{j}.b', (co,))]
            entries += _layernorm(uid, cscope, f'{p}.ln2', d)
            uid('dropout')
    entries += _layernorm(uid, scope, f'{ref}.ln', d)
    return entries


def _predictor(uid, model_scope, name, ref, d, filters, k) -> List[Entry]:
    """`StatPredictor(name=name)`: CNNDropout (convs, last_conv, then every LayerNormalization), linear."""
    cscope = f'{model_scope}/{name}/{uid("cnn_dropout")}'
    entries: List[Entry] = []
    cin = d
    convs = []
    for j, f in enumerate(filters[:-1]):
        convs.append((uid('conv1d'), j, cin, f))
        cin = f
    for _ in filters[:-1]:
        uid('activation')
    convs.append((uid('conv1d'), len(filters) - 1, cin, filters[-1]))
    uid('activation')
    for _ in filters:
        uid('dropout')
    for n, j, ci, co in convs:
        entries += [(f'{cscope}/{n}/kernel:0', f'{ref}.conv{j}.w', (k, ci, co)),
                    (f'{cscope}/{n}/bias:0', f'{ref}.conv{j}.b', (co,))]
    for j, f in enumerate(filters):
        entries += _layernorm(uid, cscope, f'{ref}.ln{j}', f)
    entries += _dense(uid, f'{model_scope}/{name}', f'{ref}.lin.w', f'{ref}.lin.b', filters[-1], 1)
    return entries


def keras_layer_table(config: dict, vocab_size: int, model_name: str = 'forward_transformer') \
        -> "List[Tuple[str, List[Entry]]]":
    """[(layer.name, [(keras variable name, package variable name, shape), ...]), ...] in `model.layers`
    order, for a ForwardTransformer built from `config` (the constructor arguments, `models.py:345-372`)."""
    c = config
    uid = _Uid()
    de, dd = c['encoder_model_dimension'], c['decoder_model_dimension']
    layers: List[Tuple[str, List[Entry]]] = []
    layers.append(('Embedding', [(f'{model_name}/Embedding/embeddings:0', 'embedding', (vocab_size, de))]))
    layers.append(('Encoder', _blocks(uid, model_name, 'Encoder', 'enc', de, list(c['encoder_num_heads']),
                                      c['encoder_dense_blocks'], c.get('encoder_feed_forward_dimension'),
                                      c.get('encoder_attention_conv_filters') or [],
                                      c.get('encoder_attention_conv_kernel'), c.get('transposed_attn_convs'))))
    layers.append(('dur_pred', _predictor(uid, model_name, 'dur_pred', 'dur', de,
                                          list(c['duration_conv_filters']), c['duration_kernel_size'])))
    layers.append(('expand', []))
    layers.append(('pitch_pred', _predictor(uid, model_name, 'pitch_pred', 'pitch', de,
                                            list(c['pitch_conv_filters']), c['pitch_kernel_size'])))
    n = uid('dense')
    layers.append((n, [(f'{model_name}/{n}/kernel:0', 'pitch_embed.w', (1, de)),
                       (f'{model_name}/{n}/bias:0', 'pitch_embed.b', (de,))]))
    layers.append(('Decoder', _blocks(uid, model_name, 'Decoder', 'dec', dd, list(c['decoder_num_heads']),
                                      c['decoder_dense_blocks'], c.get('decoder_feed_forward_dimension'),
                                      c.get('decoder_attention_conv_filters') or [],
                                      c.get('decoder_attention_conv_kernel'), c.get('transposed_attn_convs'))))
    n = uid('dense')
    layers.append((n, [(f'{model_name}/{n}/kernel:0', 'out.w', (dd, c['mel_channels'])),
                       (f'{model_name}/{n}/bias:0', 'out.b', (c['mel_channels'],))]))
    return layers


# ------------------------------------------------------------------------------ attributes, Keras style
def _save_attribute(group, name: str, values: Sequence[bytes]):
    """`save_attributes_to_hdf5_group`: one attribute, or name0, name1, ... when it would exceed the
    object header limit; an empty list is stored as an empty float64 array (numpy's default dtype)."""
    if not values:
        group.attrs[name] = np.zeros((0,), np.float64)
        return
    too_long = [v for v in values if len(v) > HDF5_OBJECT_HEADER_LIMIT]
    if too_long:
        raise RuntimeError(f'names too long for an HDF5 object header: {too_long[:2]}')
    arr = np.asarray(values, dtype='S')
    chunks = 1
    parts = np.array_split(arr, chunks)
    while any(p.nbytes > HDF5_OBJECT_HEADER_LIMIT for p in parts):
        chunks += 1
        parts = np.array_split(arr, chunks)
    if chunks == 1:
        group.attrs[name] = arr
    else:
        for i, p in enumerate(parts):
            group.attrs[f'{name}{i}'] = p


def _load_attribute(group, name: str) -> List[str]:
    """`load_attributes_from_hdf5_group`."""
    attrs = group.attrs
    dec = lambda n: n.decode('utf8') if hasattr(n, 'decode') else str(n)
    if name in attrs:
        return [dec(n) for n in np.atleast_1d(attrs[name])]
    out, i = [], 0
    while f'{name}{i}' in attrs:
        out += [dec(n) for n in np.atleast_1d(attrs[f'{name}{i}'])]
        i += 1
    return out


# ------------------------------------------------------------------------------ save / load
def save_keras_weights(path, weights: Dict[str, np.ndarray], config: dict, vocab_size: int,
                       keras_version: str = '2.4.0'):
    """Write `weights` (this package's variable names, `ForwardTransformer.weights_dict()`) as the H5 file
    Keras' `model.load_weights` expects for the reference model built from the same config."""
    table = keras_layer_table(config, vocab_size)
    w = hdf5_min.Writer()
    _save_attribute(w.root, 'layer_names', [n.encode('utf8') for n, _ in table])
    w.root.attrs['backend'] = b'tensorflow'
    w.root.attrs['keras_version'] = keras_version.encode('utf8')
    used = set()
    for lname, entries in table:
        g = w.root.create_group(lname)
        _save_attribute(g, 'weight_names', [k.encode('utf8') for k, _, _ in entries])
        for kname, ref, shape in entries:
            a = np.asarray(weights[ref], dtype=np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError(f'{ref}: shape {a.shape}, the reference model expects {shape}')
            g.create_dataset(kname, a)
            used.add(ref)
    extra = sorted(set(weights) - used)
    if extra:
        raise ValueError(f'variables with no place in the reference model: {extra[:5]}')
    w.save(str(path))


def load_keras_weights(path, config: dict, vocab_size: int) -> "OrderedDict[str, np.ndarray]":
    """Read a Keras H5 weight file of the reference model into this package's variable names.  Same
    contract as `load_weights_from_hdf5_group`: by position, counts and shapes checked."""
    table = [(n, e) for n, e in keras_layer_table(config, vocab_size) if e]
    with hdf5_min.File(str(path)) as f:
        layer_names = _load_attribute(f, 'layer_names')
        if not layer_names:
            raise ValueError(f'{path}: no layer_names attribute - not a Keras save_weights() file '
                             f'(a full-model save keeps the weights under /model_weights)')
        filtered = []
        for lname in layer_names:
            g = f[lname]
            wn = _load_attribute(g, 'weight_names')
            if wn:
                filtered.append((lname, g, wn))
        if len(filtered) != len(table):
            raise ValueError(f'You are trying to load a weight file containing {len(filtered)} layers into a '
                             f'model with {len(table)} layers.')
        out: "OrderedDict[str, np.ndarray]" = OrderedDict()
        for (lname, g, wn), (ename, entries) in zip(filtered, table):
            if len(wn) != len(entries):
                raise ValueError(f'Layer {ename!r} (file group {lname!r}) expects {len(entries)} weights, '
                                 f'but the saved weights have {len(wn)} elements.')
            for name, (kname, ref, shape) in zip(wn, entries):
                a = np.asarray(g[name])
                if tuple(a.shape) != tuple(shape):
                    raise ValueError(f'Layer {ename!r}: weight {name!r} has shape {a.shape}, the model expects '
                                     f'{shape} for {ref} ({kname}).')
                out[ref] = np.ascontiguousarray(a, dtype=np.float32).reshape(shape)
    return out
