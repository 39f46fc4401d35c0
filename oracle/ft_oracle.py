"""CPU restatement ORACLE of the ForwardTransformer hot path of as-ideas/TransformerTTS.

TEST INFRASTRUCTURE ONLY.  Nothing under ``transformertts_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only
as the checker / the timed CPU baseline - never as the thing shipped.

PARITY UNPINNED: TensorFlow (>=2.2, the reference's arithmetic backend, requirements.txt:7) is not
installable in the build image and the reference's own tests pin nothing on this path
(SURVEY.md section 8c).  The only reference-authored known answer on the path is the ``Expand``
docstring example (model/layers.py:532-542), which tests/test_oracle.py checks.  Everything else
here is a line-by-line restatement of the reference *source*, with the Keras defaults it relies on
restated from the public Keras documentation (marked [3P]).
What IS pinned against the reference's own code: its plain-NumPy / pure-Python pieces run in the build
container (tests/golden/make_reference_fixtures.py imports them from /root/reference with empty stand-ins
for the absent third-party modules) - positional_encoding, the two padding-mask constructors (over a
five-op NumPy stand-in for TF), the tokenizer, the lr / reduction schedules; tests/test_reference_fixtures.py
holds this oracle (and the product's host code) to those vectors bit for bit.  Beyond that the reference's
MODEL SOURCE itself (model/models.py, model/layers.py, utils/losses.py) is executed over a torch-float64 stand-in
for the ~45 TensorFlow names it uses (tests/_tf_shim.py, tests/golden/make_reference_source_run.py): forward
outputs, losses, the gradient of every variable and predict() agree with this oracle to 1e-10 relative
(tests/test_reference_source_run.py) - the wiring is pinned, TensorFlow's floating point and its Adam are not.

The restatement is written once in torch-CPU and parameterised by dtype:
  * ``torch.float64``  - the truth the 1e-4 relative tolerance is measured against;
  * ``torch.float32``  - the same graph in the reference's own precision; this is what bench.py
                         times as the ``cpu_baseline`` (kind "port").
torch autograd supplies reference gradients; the TF-form Adam is restated by hand.

All file:line citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-6          # model/layers.py:27,64,96,207,295,508  (LayerNormalization(epsilon=1e-6))
VOCAB_SIZE = 127       # data/text/symbols.py:1-12 -> 126 symbols + pad (data/text/tokenizer.py:19)


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
def make_config(d_model=256, enc_heads=(4,) * 6, dec_heads=(4,) * 6, ffn=1024,
                enc_dense_blocks=None, dec_dense_blocks=None, conv_filters=None, conv_kernel=3,
                dur_filters=(256, 226), pitch_filters=(256, 226), dur_kernel=3, pitch_kernel=3,
                mel_channels=80, enc_max_pos=2000, dec_max_pos=10000, dropout_rate=0.0,
                predictors_dropout=0.0) -> dict:
    """Flat config dict with the reference's constructor keyword names (model/models.py:345-372)."""
    enc_heads, dec_heads = list(enc_heads), list(dec_heads)
    return dict(
        encoder_model_dimension=d_model, decoder_model_dimension=d_model,
        dropout_rate=dropout_rate, decoder_num_heads=dec_heads, encoder_num_heads=enc_heads,
        encoder_max_position_encoding=enc_max_pos, decoder_max_position_encoding=dec_max_pos,
        encoder_dense_blocks=len(enc_heads) if enc_dense_blocks is None else enc_dense_blocks,
        decoder_dense_blocks=len(dec_heads) if dec_dense_blocks is None else dec_dense_blocks,
        duration_conv_filters=list(dur_filters), pitch_conv_filters=list(pitch_filters),
        duration_kernel_size=dur_kernel, pitch_kernel_size=pitch_kernel,
        predictors_dropout=predictors_dropout, mel_channels=mel_channels,
        phoneme_language='en-us', with_stress=True, model_breathing=False,
        transposed_attn_convs=True,
        encoder_attention_conv_filters=None if conv_filters is None else list(conv_filters),
        decoder_attention_conv_filters=None if conv_filters is None else list(conv_filters),
        encoder_attention_conv_kernel=conv_kernel, decoder_attention_conv_kernel=conv_kernel,
        encoder_feed_forward_dimension=ffn, decoder_feed_forward_dimension=ffn,
    )


def tiny_config(**kw) -> dict:
    """BASELINE.json configs[0]: d_model=64, 2+2 layers, 80-bin mel."""
    base = dict(d_model=64, enc_heads=(2, 2), dec_heads=(2, 2), ffn=256,
                dur_filters=(64, 64), pitch_filters=(64, 64))
    base.update(kw)
    return make_config(**base)


# ----------------------------------------------------------------------------------------------
# weights: names, shapes, Keras-default init  [3P]
# ----------------------------------------------------------------------------------------------
def _blocks_spec(prefix: str, d: int, heads: List[int], dense_blocks: int, ffn, conv_filters,
                 conv_kernel) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    s[f'{prefix}.ln.gamma'] = (d,)                      # layers.py:295
    s[f'{prefix}.ln.beta'] = (d,)
    s[f'{prefix}.pos_scalar'] = ()                      # layers.py:282  tf.Variable(1.)
    for i, _ in enumerate(heads):
        p = f'{prefix}.blk{i}'
        for n in ('wq', 'wk', 'wv'):                    # layers.py:116-118
            s[f'{p}.{n}'] = (d, d)
            s[f'{p}.b{n[1]}'] = (d,)
        s[f'{p}.wo'] = (2 * d, d)                       # layers.py:120,148-149 concat([q_in, ctx])
        s[f'{p}.bo'] = (d,)
        s[f'{p}.ln1.gamma'] = (d,)                      # layers.py:207
        s[f'{p}.ln1.beta'] = (d,)
        if i < dense_blocks:                            # layers.py:285-288 SelfAttentionDenseBlock
            s[f'{p}.ffn.w1'] = (d, ffn)                 # layers.py:93
            s[f'{p}.ffn.b1'] = (ffn,)
            s[f'{p}.ffn.w2'] = (ffn, d)                 # layers.py:94
            s[f'{p}.ffn.b2'] = (d,)
        else:                                           # layers.py:289-294 SelfAttentionConvBlock
            cin = d
            for j, f in enumerate(conv_filters):        # layers.py:19-26  Conv1D kernel [k,in,out]
                s[f'{p}.conv{j}.w'] = (conv_kernel, cin, f)
                s[f'{p}.conv{j}.b'] = (f,)
                cin = f
        s[f'{p}.ln2.gamma'] = (d,)                      # layers.py:96 / :27
        s[f'{p}.ln2.beta'] = (d,)
    return s


def _predictor_spec(prefix: str, d: int, filters: List[int], k: int) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    cin = d
    for j, f in enumerate(filters):                     # layers.py:498-508
        s[f'{prefix}.conv{j}.w'] = (k, cin, f)
        s[f'{prefix}.conv{j}.b'] = (f,)
        s[f'{prefix}.ln{j}.gamma'] = (f,)
        s[f'{prefix}.ln{j}.beta'] = (f,)
        cin = f
    s[f'{prefix}.lin.w'] = (cin, 1)                     # layers.py:479
    s[f'{prefix}.lin.b'] = (1,)
    return s


def weight_spec(cfg: dict, vocab_size: int = VOCAB_SIZE) -> "OrderedDict[str, tuple]":
    """Ordered name -> shape map of every trainable variable (model/models.py:381-422)."""
    de, dd = cfg['encoder_model_dimension'], cfg['decoder_model_dimension']
    s = OrderedDict()
    s['embedding'] = (vocab_size, de)                   # models.py:381-383
    s.update(_blocks_spec('enc', de, cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                          cfg['encoder_feed_forward_dimension'],
                          cfg['encoder_attention_conv_filters'], cfg['encoder_attention_conv_kernel']))
    s.update(_predictor_spec('dur', de, cfg['duration_conv_filters'], cfg['duration_kernel_size']))
    s.update(_predictor_spec('pitch', de, cfg['pitch_conv_filters'], cfg['pitch_kernel_size']))
    s['pitch_embed.w'] = (1, de)                        # models.py:410
    s['pitch_embed.b'] = (de,)
    s.update(_blocks_spec('dec', dd, cfg['decoder_num_heads'], cfg['decoder_dense_blocks'],
                          cfg['decoder_feed_forward_dimension'],
                          cfg['decoder_attention_conv_filters'], cfg['decoder_attention_conv_kernel']))
    s['out.w'] = (dd, cfg['mel_channels'])              # models.py:422
    s['out.b'] = (cfg['mel_channels'],)
    return s


def init_weights(cfg: dict, seed: int = 0, vocab_size: int = VOCAB_SIZE,
                 perturb: float = 0.0) -> "OrderedDict[str, np.ndarray]":
    """Keras default initialisers [3P]: Dense/Conv1D glorot_uniform kernels (limit
    sqrt(6/(fan_in+fan_out)), conv fans multiplied by the receptive field) + zero bias; Embedding
    uniform(-0.05, 0.05); LayerNormalization gamma=1, beta=0; pos_encoding_scalar = 1.
    ``perturb`` > 0 adds N(0, perturb^2) noise to biases/LN params/pos scalars so that parity tests
    exercise them (a freshly initialised model has them at exactly 0/1)."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in weight_spec(cfg, vocab_size).items():
        leaf = name.split('.')[-1]
        if name == 'embedding':
            a = rng.uniform(-0.05, 0.05, size=shape)
        elif leaf == 'gamma' or leaf == 'pos_scalar':
            a = np.ones(shape) + perturb * rng.standard_normal(shape)
        elif leaf == 'beta' or (leaf.startswith('b') and len(shape) == 1):
            a = np.zeros(shape) + perturb * rng.standard_normal(shape)
        elif len(shape) == 2:
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            a = rng.uniform(-lim, lim, size=shape)
        elif len(shape) == 3:
            k, cin, cout = shape
            lim = math.sqrt(6.0 / (k * cin + k * cout))
            a = rng.uniform(-lim, lim, size=shape)
        else:
            raise ValueError(name)
        w[name] = np.asarray(a, dtype=np.float32).astype(np.float64)  # fp32-representable values
    return w


# ----------------------------------------------------------------------------------------------
# model/transformer_utils.py
# ----------------------------------------------------------------------------------------------
def positional_encoding(position: int, model_dim: int) -> np.ndarray:
    """transformer_utils.py:5-21.  angle = pos / 10000^(2*(i//2)/float32(model_dim)), computed in
    numpy float64 (np.power(10000, float64 / float32) -> float64), sin on even columns, cos on odd
    columns, then tf.cast(..., float32)."""
    pos = np.arange(position)[:, np.newaxis]
    i = np.arange(model_dim)[np.newaxis, :]
    angle_rates = 1 / np.power(10000, (2 * (i // 2)) / np.float32(model_dim))
    angle_rads = pos * angle_rates
    angle_rads[:, 0::2] = np.sin(angle_rads[:, 0::2])
    angle_rads[:, 1::2] = np.cos(angle_rads[:, 1::2])
    return angle_rads.astype(np.float32)   # [position, model_dim]; the leading 1-axis is implicit


def create_encoder_padding_mask(seq: torch.Tensor, dtype) -> torch.Tensor:
    """transformer_utils.py:24-26: float(seq == 0)[:, None, None, :]."""
    return (seq == 0).to(dtype)[:, None, None, :]


def create_mel_padding_mask(seq: torch.Tensor) -> torch.Tensor:
    """transformer_utils.py:29-32: float(sum_c |x| == 0)[:, None, None, :]  (content-derived)."""
    s = seq.abs().sum(-1)
    return (s == 0).to(seq.dtype)[:, None, None, :]


# ----------------------------------------------------------------------------------------------
# model/layers.py
# ----------------------------------------------------------------------------------------------
def layer_norm(x, gamma, beta):
    """Keras LayerNormalization(epsilon=1e-6) over the last axis: biased variance, eps inside the
    sqrt [3P]."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * gamma + beta


def conv1d_same(x, w, b):
    """Keras Conv1D(padding='same', stride 1) on channels-last [B,T,Cin] with kernel [k,Cin,Cout]:
    cross-correlation, zero padding (k-1)//2 left, k//2 right [3P]."""
    k = w.shape[0]
    xp = F.pad(x, (0, 0, (k - 1) // 2, k // 2))
    T = x.shape[1]
    out = b
    for j in range(k):
        out = out + xp[:, j:j + T, :] @ w[j]
    return out


class _Dropout:
    """Keras inverted dropout x * keep / (1 - p) [3P].  The oracle only needs rate 0 (parity runs)
    and a seeded torch stream (statistical tests / CPU-baseline timing with the real work)."""

    def __init__(self, seed: int = 0):
        self.gen = torch.Generator().manual_seed(seed)

    def __call__(self, x, rate: float, training: bool):
        if not training or rate == 0.0:
            return x
        keep = (torch.rand(x.shape, generator=self.gen) >= rate).to(x.dtype)
        return x * keep / (1.0 - rate)


def scaled_dot_product_attention(q, k, v, mask, rate, training, drop):
    """layers.py:176-195."""
    logits = q @ k.transpose(-1, -2)
    dk = torch.tensor(float(k.shape[-1]), dtype=q.dtype)
    logits = logits / torch.sqrt(dk)
    if mask is not None:
        logits = logits + mask * -1e9
    weights = torch.softmax(logits, dim=-1)
    weights = drop(weights, rate, training)            # returned weights are post-dropout
    return weights @ v, weights


def multi_head_attention(W, p, x, mask, heads, rate, training, drop):
    """layers.py:131-151 with v = k = q_in = x (SelfAttentionResNorm, layers.py:210)."""
    B, T, d = x.shape
    depth = d // heads

    def split(t):                                      # layers.py:123-129
        return t.reshape(B, T, heads, depth).permute(0, 2, 1, 3)

    q = split(x @ W[f'{p}.wq'] + W[f'{p}.bq'])
    k = split(x @ W[f'{p}.wk'] + W[f'{p}.bk'])
    v = split(x @ W[f'{p}.wv'] + W[f'{p}.bv'])
    ctx, weights = scaled_dot_product_attention(q, k, v, mask, rate, training, drop)
    ctx = ctx.permute(0, 2, 1, 3).reshape(B, T, d)     # layers.py:144-147
    concat_query = torch.cat([x, ctx], dim=-1)         # layers.py:148
    out = concat_query @ W[f'{p}.wo'] + W[f'{p}.bo']   # layers.py:149
    return drop(out, rate, training), weights          # layers.py:150


def self_attention_block(W, p, x, mask, heads, dense, rate, training, drop):
    """SelfAttentionDenseBlock layers.py:226-230 / SelfAttentionConvBlock layers.py:259-264.
    TransposedCNNResNorm transposes with the identity permutation (layers.py:74,77), so it equals
    CNNResNorm."""
    attn_out, weights = multi_head_attention(W, p, x, mask, heads, rate, training, drop)
    a = layer_norm(attn_out + x, W[f'{p}.ln1.gamma'], W[f'{p}.ln1.beta'])      # layers.py:211
    dense_mask = 1.0 - mask[:, 0, 0, :, None]                                   # layers.py:228
    a = a * dense_mask
    if dense:
        h = torch.relu(a @ W[f'{p}.ffn.w1'] + W[f'{p}.ffn.b1'])                 # layers.py:99
        h = h @ W[f'{p}.ffn.w2'] + W[f'{p}.ffn.b2']                             # layers.py:100
    else:
        h = a
        j = 0
        while f'{p}.conv{j}.w' in W:
            last = f'{p}.conv{j + 1}.w' not in W
            h = conv1d_same(h, W[f'{p}.conv{j}.w'], W[f'{p}.conv{j}.b'])        # layers.py:30-38
            if not last:
                h = torch.relu(h)
            j += 1
    h = drop(h, rate, training)
    out = layer_norm(h + a, W[f'{p}.ln2.gamma'], W[f'{p}.ln2.beta'])            # layers.py:102 / :40
    return out * dense_mask, weights


def self_attention_blocks(W, prefix, name, x, mask, heads_list, dense_blocks, pe, rate, training,
                          drop, taps=None):
    """SelfAttentionBlocks.call layers.py:297-310.  taps (test instrumentation): a list that receives
    (f'{prefix}.blk{i}', block output) for every block - used to report error per layer depth."""
    T = x.shape[1]
    x = layer_norm(x, W[f'{prefix}.ln.gamma'], W[f'{prefix}.ln.beta'])          # layers.py:299
    x = x + W[f'{prefix}.pos_scalar'] * pe[None, :T, :]                         # layers.py:300
    x = drop(x, rate, training)                                                 # layers.py:301
    attn = OrderedDict()
    for i, h in enumerate(heads_list):
        dense = i < dense_blocks
        x, w = self_attention_block(W, f'{prefix}.blk{i}', x, mask, h, dense, rate, training, drop)
        if dense:
            attn[f'{name}_DenseBlock{i + 1}_SelfAttention'] = w                 # layers.py:305
        else:
            attn[f'{name}_ConvBlock{i - dense_blocks + 1}_SelfAttention'] = w   # layers.py:308
        if taps is not None:
            taps.append((f'{prefix}.blk{i}', x.detach()))
    return x, attn


def stat_predictor(W, p, x, mask, n_layers, relu_head, rate, training, drop):
    """StatPredictor.call layers.py:481-485 + CNNDropout.call layers.py:510-524:
    x*mask -> [Conv1D same -> relu -> LN -> dropout] x n -> Dense(1, act) -> *mask."""
    x = x * mask
    for j in range(n_layers):
        x = conv1d_same(x, W[f'{p}.conv{j}.w'], W[f'{p}.conv{j}.b'])
        x = torch.relu(x)
        x = layer_norm(x, W[f'{p}.ln{j}.gamma'], W[f'{p}.ln{j}.beta'])
        x = drop(x, rate, training)
    x = x @ W[f'{p}.lin.w'] + W[f'{p}.lin.b']
    if relu_head:
        x = torch.relu(x)
    return x * mask


def round_half_even(x: np.ndarray) -> np.ndarray:
    """tf.math.round rounds half to even [3P]; numpy's rint does the same."""
    return np.rint(x)


def expand_literal_np(x: np.ndarray, dimensions: np.ndarray) -> np.ndarray:
    """Expand.call layers.py:549-565 restated operation by operation in numpy (tile + boolean
    mask + ragged re-assembly).  x [B,T,C]; dimensions [B,T,1] (int or float)."""
    dims = dimensions[..., 0]
    dims = round_half_even(dims).astype(np.int32)                      # :551
    B, T, C = x.shape
    max_dim = int(dims.max()) if dims.size else 0                      # :555
    if (dims < 0).any():
        raise ValueError('negative duration (tf.RaggedTensor.from_row_lengths would raise)')
    flat = dims.reshape(-1)                                            # :557 row_lengths
    index_masks = np.zeros((flat.size, max_dim))
    for r, n in enumerate(flat):
        index_masks[r, :n] = 1.0                                       # ragged ones -> to_tensor()
    index_masks = index_masks.reshape(B, T * max_dim).astype(np.float32)   # :558
    non_zeros = T * max_dim - (max_dim - dims).sum(axis=1)             # :559
    tiled = np.tile(x, (1, 1, max_dim))                                # :561
    reshaped = tiled.reshape(B, T * max_dim, C)                        # :562
    mask_reshape = reshaped * index_masks[:, :, None]                  # :563
    rows = mask_reshape[index_masks > 0]                               # :564 boolean_mask, row-major
    out_len = int(non_zeros.max()) if B else 0
    out = np.zeros((B, out_len, C), dtype=x.dtype)
    o = 0
    for b in range(B):
        n = int(non_zeros[b])
        out[b, :n] = rows[o:o + n]
        o += n
    return out                                                         # :565 ragged.to_tensor()


def expand_indices_np(dimensions: np.ndarray):
    """The index formulation of Expand the HIP kernel implements: int32 index table
    idx[b, j] = phoneme whose cumulative-duration interval contains frame j (or -1 for padding),
    per-sample lengths, and the padded output length.  Bit-exact target."""
    dims = round_half_even(dimensions[..., 0]).astype(np.int32)
    B, T = dims.shape
    lens = dims.sum(axis=1).astype(np.int32)
    out_len = int(lens.max()) if B else 0
    idx = -np.ones((B, out_len), dtype=np.int32)
    for b in range(B):
        idx[b, :lens[b]] = np.repeat(np.arange(T, dtype=np.int32), dims[b])
    return idx, lens, out_len


def expand_torch(x: torch.Tensor, dimensions: torch.Tensor) -> torch.Tensor:
    """Differentiable Expand (gather by the index table) used inside the torch graph; the literal
    numpy restatement above is checked equal to it in tests/test_oracle.py."""
    idx, lens, out_len = expand_indices_np(dimensions.detach().cpu().numpy())
    B, T, C = x.shape
    idx_t = torch.from_numpy(idx.astype(np.int64))
    valid = (idx_t >= 0)
    g = torch.gather(x, 1, idx_t.clamp(min=0)[:, :, None].expand(B, out_len, C))
    return g * valid[:, :, None].to(x.dtype)


# ----------------------------------------------------------------------------------------------
# utils/losses.py
# ----------------------------------------------------------------------------------------------
def masked_mean_absolute_error(targets, pred):
    """utils/losses.py:41-49 with mask=None (the only way weighted_sum_losses calls it): Keras
    MeanAbsoluteError = mean over every element; integer targets are cast to float [3P]."""
    return (targets.to(pred.dtype) - pred).abs().mean()


def keras_weighted_loss_mean(per_sample, sample_weight=None):
    """The reduction every Keras loss OBJECT of utils/losses.py applies (`Loss.__call__`, reduction
    SUM_OVER_BATCH_SIZE [3P]): the per-sample losses (already averaged over the last axis) are multiplied by
    sample_weight and their SUM is divided by the NUMBER of per-sample losses - not by the sum of the weights.  With
    sample_weight=None (how the ForwardTransformer's MAE losses are called, utils/losses.py:41-49) that is the plain
    mean above; the weighted form is pinned by the reference's own known answers (tests/test_loss.py:10-26, the
    crossentropy losses of the same file), which fix the convention rather than leave it to memory."""
    per_sample = per_sample if sample_weight is None else per_sample * sample_weight.to(per_sample.dtype)
    return per_sample.sum() / per_sample.numel()


def masked_crossentropy(targets, logits, index=None, scaling=1.0):
    """utils/losses.py:4-29 (not on the ForwardTransformer path - restated only because the reference's tests pin the
    Keras reduction with it): sparse categorical crossentropy from logits, weight 0 at padding targets (id 0), weight
    `scaling` at targets == index."""
    logp = torch.log_softmax(logits.to(torch.float64), dim=-1)
    ce = -torch.gather(logp, -1, targets.long()[..., None])[..., 0]
    w = (targets != 0).to(torch.float64)
    if index is not None:
        w = w + (targets == index).to(torch.float64) * (scaling - 1.0)
    return keras_weighted_loss_mean(ce, w)


def weighted_sum_losses(targets, pred, coeffs):
    """utils/losses.py:63-70."""
    total = 0
    vals = []
    for t, p, c in zip(targets, pred, coeffs):
        l = masked_mean_absolute_error(t, p)
        vals.append(l)
        total = total + c * l
    return total, vals


LOSS_WEIGHTS = [1., 1., 3.]      # model/models.py:485


# ----------------------------------------------------------------------------------------------
# model/models.py  ForwardTransformer
# ----------------------------------------------------------------------------------------------
class ForwardTransformerOracle:
    def __init__(self, cfg: dict, weights: Dict[str, np.ndarray], dtype=torch.float64,
                 dropout_seed: int = 0):
        self.cfg = cfg
        self.dtype = dtype
        self.W = OrderedDict((k, torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True))
                             for k, v in weights.items())
        self.pe_enc = torch.tensor(positional_encoding(cfg['encoder_max_position_encoding'],
                                                       cfg['encoder_model_dimension']), dtype=dtype)
        self.pe_dec = torch.tensor(positional_encoding(cfg['decoder_max_position_encoding'],
                                                       cfg['decoder_model_dimension']), dtype=dtype)
        self.drop = _Dropout(dropout_seed)
        # Adam state (utils/training_config_manager.py:102-106)
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in self.W.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in self.W.items())
        self.iterations = 0
        self.learning_rate = 1e-4

    # model/models.py:518-550
    def call(self, x, target_durations=None, target_pitch=None, training=False,
             durations_scalar=1., max_durations_mask=None, min_durations_mask=None):
        cfg, W, dt = self.cfg, self.W, self.dtype
        rate, prate = cfg['dropout_rate'], cfg['predictors_dropout']
        x = torch.as_tensor(x).long()
        encoder_padding_mask = create_encoder_padding_mask(x, dt)                      # :521
        h = W['embedding'][x]                                                          # :522
        h, enc_attn = self_attention_blocks(W, 'enc', 'Encoder', h, encoder_padding_mask,
                                            cfg['encoder_num_heads'], cfg['encoder_dense_blocks'],
                                            self.pe_enc, rate, training, self.drop,
                                            getattr(self, 'taps', None))               # :523
        padding_mask = 1. - encoder_padding_mask[:, 0, 0, :, None]                     # :524
        durations = stat_predictor(W, 'dur', h, padding_mask, len(cfg['duration_conv_filters']),
                                   True, prate, training, self.drop)                   # :525
        pitch = stat_predictor(W, 'pitch', h, padding_mask, len(cfg['pitch_conv_filters']),
                               False, prate, training, self.drop)                      # :526
        p_in = torch.as_tensor(target_pitch).to(dt) if target_pitch is not None else pitch
        pitch_embed = torch.relu(p_in @ W['pitch_embed.w'] + W['pitch_embed.b'])       # :527-530
        h = h + pitch_embed                                                            # :531
        if target_durations is not None:
            use_durations = torch.as_tensor(target_durations)                          # :533
        else:
            use_durations = durations * durations_scalar                               # :535
        if max_durations_mask is not None:                                             # :536-537
            use_durations = torch.minimum(use_durations.to(dt),
                                          torch.as_tensor(max_durations_mask).to(dt)[..., None])
        if min_durations_mask is not None:                                             # :538-539
            use_durations = torch.maximum(use_durations.to(dt),
                                          torch.as_tensor(min_durations_mask).to(dt)[..., None])
        if use_durations.dtype in (torch.float32, torch.float64):
            # the reference rounds an fp32 tensor; round what fp32 would see
            use_durations = use_durations.detach().to(torch.float32)
        mels = expand_torch(h, use_durations)                                          # :540
        expanded_mask = create_mel_padding_mask(mels.detach())                         # :541
        mels, dec_attn = self_attention_blocks(W, 'dec', 'Decoder', mels, expanded_mask,
                                               cfg['decoder_num_heads'], cfg['decoder_dense_blocks'],
                                               self.pe_dec, rate, training, self.drop,
                                               getattr(self, 'taps', None))            # :542
        mels = mels @ W['out.w'] + W['out.b']                                          # :543
        return {'mel': mels, 'duration': durations, 'pitch': pitch,
                'expanded_mask': expanded_mask, 'encoder_attention': enc_attn,
                'decoder_attention': dec_attn}                                         # :544-549

    def _losses(self, out, target_sequence, target_durations, target_pitch):
        mel_len = int(target_sequence.shape[1])                                        # :467
        return weighted_sum_losses((target_sequence, target_durations, target_pitch),
                                   (out['mel'][:, :mel_len, :], out['duration'], out['pitch']),
                                   LOSS_WEIGHTS)                                       # :470-477

    # model/models.py:492-507
    def val_step(self, input_sequence, target_sequence, target_durations, target_pitch):
        td = torch.as_tensor(target_durations)[..., None]                              # :493
        tp = torch.as_tensor(target_pitch).to(self.dtype)[..., None]                   # :494
        ts = torch.as_tensor(target_sequence).to(self.dtype)
        with torch.no_grad():
            out = self.call(input_sequence, td, target_pitch=tp, training=False)
            loss, vals = self._losses(out, ts, td, tp)
        out.update({'loss': loss, 'losses': {'mel': vals[0], 'duration': vals[1], 'pitch': vals[2]}})
        return out

    # model/models.py:464-482
    def train_step(self, input_sequence, target_sequence, target_durations, target_pitch,
                   apply: bool = True):
        td = torch.as_tensor(target_durations)[..., None]                              # :465
        tp = torch.as_tensor(target_pitch).to(self.dtype)[..., None]                   # :466
        ts = torch.as_tensor(target_sequence).to(self.dtype)
        for w in self.W.values():
            w.grad = None
        out = self.call(input_sequence, td, target_pitch=tp, training=True)            # :469
        loss, vals = self._losses(out, ts, td, tp)
        loss.backward()                                                                # :480
        grads = OrderedDict((k, (w.grad if w.grad is not None else torch.zeros_like(w)).clone())
                            for k, w in self.W.items())
        if apply:
            self.apply_gradients(grads)                                                # :481
        out = {k: (v.detach() if torch.is_tensor(v) else
                   OrderedDict((kk, vv.detach()) for kk, vv in v.items())) for k, v in out.items()}
        out.update({'loss': loss.detach(),
                    'losses': {'mel': vals[0].detach(), 'duration': vals[1].detach(),
                               'pitch': vals[2].detach()},
                    'grads': grads})
        return out

    def apply_gradients(self, grads, beta_1=0.9, beta_2=0.98, epsilon=1e-9):
        """tf.keras.optimizers.Adam (non-amsgrad) [3P], hyper-parameters from
        utils/training_config_manager.py:102-106:
            t += 1; lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
            m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  theta -= lr_t * m / (sqrt(v) + eps)
        (epsilon OUTSIDE the bias-corrected sqrt - differs from torch.optim.Adam)."""
        self.iterations += 1
        t = self.iterations
        lr_t = self.learning_rate * math.sqrt(1 - beta_2 ** t) / (1 - beta_1 ** t)
        with torch.no_grad():
            for k, w in self.W.items():
                g = grads[k]
                self.m[k].mul_(beta_1).add_(g, alpha=1 - beta_1)
                self.v[k].mul_(beta_2).addcmul_(g, g, value=1 - beta_2)
                w.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + epsilon))

    @property
    def step(self):                                                                    # :514-516
        return self.iterations

    def weights_numpy(self):
        return OrderedDict((k, v.detach().numpy().copy()) for k, v in self.W.items())


# ----------------------------------------------------------------------------------------------
# synthetic inputs  
# ----------------------------------------------------------------------------------------------
def synthetic_batch(B: int, Tp: int, Tm: int, mel_channels: int = 80, seed: int = 1234,
                    ragged: bool = False, vocab_size: int = VOCAB_SIZE):
    """max-shape set (ragged=False): every sample Tp phonemes / Tm frames, durations =
    multinomial(Tm, uniform) (zeros allowed, sum == Tm), pitch ~ N(0,1) with 30 % exact zeros, mel ~
    clip(N(-5,2), -11.5129, 2).  ragged=True: LJ-dist set - per-sample lengths, zero padded, at
    least one sample at each maximum, sum(dur_b) == mel_len_b."""
    rng = np.random.default_rng(seed)
    tokens = np.zeros((B, Tp), dtype=np.int32)
    durs = np.zeros((B, Tp), dtype=np.int32)
    pitch = np.zeros((B, Tp), dtype=np.float32)
    mel = np.zeros((B, Tm, mel_channels), dtype=np.float32)
    for b in range(B):
        if ragged and b > 0:
            tp = int(rng.integers(max(1, (3 * Tp) // 10), Tp + 1))
            tm = int(min(Tm, max(tp, rng.integers(max(1, (3 * Tm) // 10), Tm + 1))))
        else:
            tp, tm = Tp, Tm
        tokens[b, :tp] = rng.integers(1, vocab_size, size=tp)
        durs[b, :tp] = rng.multinomial(tm, np.full(tp, 1.0 / tp))
        p = rng.standard_normal(tp).astype(np.float32)
        p[rng.random(tp) < 0.3] = 0.0
        pitch[b, :tp] = p
        mel[b, :tm] = np.clip(rng.normal(-5.0, 2.0, size=(tm, mel_channels)), -11.5129, 2.0)
    return tokens, mel, durs, pitch
