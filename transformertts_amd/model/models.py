"""ForwardTransformer - MI355X-native mirror of reference model/models.py:344-642.

Same constructor keywords, same methods (from_config, train_step, val_step, call/__call__, predict,
set_constants, step, _compile, save_model, load_model) and the same model_out dictionary keys as
the reference, so train_tts.py / predict_tts.py call sites keep working.  Underneath there is no
TensorFlow graph: every layer is a fused HIP kernel of libttsmi.so driven through torch autograd
Functions (transformertts_amd/ops.py); parameters, gradients and Adam state live in three flat fp32
buffers (one fused TF-form Adam launch, one RCCL all-reduce per step when data-parallel).

Differences a caller can observe (all documented in INTEGRATION.md):
  * tensors are torch CUDA tensors instead of tf.Tensors;
  * the 12 attention maps [B,H,T,T] the reference returns from every call are only materialised
    when ``model.return_attention`` is True (default: True for val_step/predict/call, False for
    train_step) - writing them is 5 GB of HBM traffic per step at the LJSpeech batch shape;
  * the decoder padding mask is derived from the summed durations instead of the content test
    ``sum|x| == 0`` (transformer_utils.py:29-32); identical unless an expanded row is exactly zero.
"""
from __future__ import annotations

import math
import os
import subprocess
from collections import OrderedDict
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch
import yaml

from .. import ops
from ..data.text import TextToTokens
from ..utils.losses import masked_mean_absolute_error, weighted_sum_losses
from .transformer_utils import positional_encoding


_DENSE_STACK = os.environ.get('TTSMI_DENSE_STACK', '1') != '0'      # A/B knob: 0 = one autograd node per planned block
_PRED_LATE = os.environ.get('TTSMI_PRED_LATE', '1') != '0'      # A/B knob: 0 = predictors issued before the decoder (round 2)
_PRED_ONE_NODE = os.environ.get('TTSMI_PRED_ONE_NODE', '1') != '0'  # A/B knob: 0 = eight autograd nodes per StatPredictor
# A/B knob: 0 = the per-layer path (conv blocks) leaves the sums of multiply-used tensors' gradients to autograd
_GRAD_SINK = os.environ.get('TTSMI_GRAD_SINK', '1') == '1'
_DROPBITS_CONV = os.environ.get('TTSMI_ATTN_DROPBITS_CONV', '0') == '1'
_CHAIN_PREPACK = os.environ.get('TTSMI_CHAIN_PREPACK', '1') != '0'      # 0: the chain kernels' weight streams are packed in line (A/B knob)


def _on_device(fn):
    """Run a method with the model's GPU current: `device=` decides where the tensors live, but the launch
    stream, the weight-gradient side stream and every temporary follow torch's CURRENT device."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kw):
        # (the GEMM family of the fp32-tensor path this model's launches take: exact fp32, or three bf16 products)
        ops.set_f32_gemm_dtype(ops._lib.TTSMI_BF16X3 if getattr(self, 'precision', 'f32') == 'bf16x3' else ops.TTSMI_F32)
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *args, **kw)
        with torch.cuda.device(self.device):
            return fn(self, *args, **kw)
    return wrapper


def _blocks_spec(prefix, d, heads, dense_blocks, ffn, conv_filters, conv_kernel):
    s = OrderedDict()
    s[f'{prefix}.ln.gamma'] = (d,)
    s[f'{prefix}.ln.beta'] = (d,)
    s[f'{prefix}.pos_scalar'] = ()
    for i, _ in enumerate(heads):
        p = f'{prefix}.blk{i}'
        s[f'{p}.wqkv'] = (d, 3 * d)          # Wq | Wk | Wv fused (layers.py:116-118)
        s[f'{p}.bqkv'] = (3 * d,)
        s[f'{p}.wo'] = (2 * d, d)            # Dense(concat([q_in, ctx])) (layers.py:148-149)
        s[f'{p}.bo'] = (d,)
        s[f'{p}.ln1.gamma'] = (d,)
        s[f'{p}.ln1.beta'] = (d,)
        if i < dense_blocks:
            s[f'{p}.ffn.w1'] = (d, ffn)
            s[f'{p}.ffn.b1'] = (ffn,)
            s[f'{p}.ffn.w2'] = (ffn, d)
            s[f'{p}.ffn.b2'] = (d,)
        else:
            cin = d
            for j, f in enumerate(conv_filters):
                s[f'{p}.conv{j}.w'] = (conv_kernel, cin, f)
                s[f'{p}.conv{j}.b'] = (f,)
                cin = f
        s[f'{p}.ln2.gamma'] = (d,)
        s[f'{p}.ln2.beta'] = (d,)
    return s


def _predictor_spec(prefix, d, filters, k):
    s = OrderedDict()
    cin = d
    for j, f in enumerate(filters):
        s[f'{prefix}.conv{j}.w'] = (k, cin, f)
        s[f'{prefix}.conv{j}.b'] = (f,)
        s[f'{prefix}.ln{j}.gamma'] = (f,)
        s[f'{prefix}.ln{j}.beta'] = (f,)
        cin = f
    s[f'{prefix}.lin.w'] = (cin, 1)
    s[f'{prefix}.lin.b'] = (1,)
    return s


class FlatParams:
    """All trainable variables as views of ONE flat fp32 buffer (+ matching gradient / Adam m / v
    buffers).  Every tensor starts on a 16-byte boundary so the float4 kernel paths apply."""

    ALIGN = 8     # 32 B in fp32, 16 B in the flat bf16 shadow copy (views of it are GEMM operands)

    def __init__(self, spec: "OrderedDict[str, tuple]", device):
        self.spec = spec
        self.offsets = OrderedDict()
        off = 0
        for name, shape in spec.items():
            n = int(np.prod(shape)) if len(shape) else 1
            self.offsets[name] = (off, n)
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.w: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        for name, shape in spec.items():
            o, n = self.offsets[name]
            self.w[name] = self.data[o:o + n].view(shape).requires_grad_()
            self.g[name] = self.grad[o:o + n].view(shape)

    @property
    def n_params(self) -> int:
        return sum(n for _, n in self.offsets.values())


class ForwardTransformer:
    def __init__(self,
                 encoder_model_dimension: int,
                 decoder_model_dimension: int,
                 dropout_rate: float,
                 decoder_num_heads: list,
                 encoder_num_heads: list,
                 encoder_max_position_encoding: int,
                 decoder_max_position_encoding: int,
                 encoder_dense_blocks: int,
                 decoder_dense_blocks: int,
                 duration_conv_filters: list,
                 pitch_conv_filters: list,
                 duration_kernel_size: int,
                 pitch_kernel_size: int,
                 predictors_dropout: float,
                 mel_channels: int,
                 phoneme_language: str,
                 with_stress: bool,
                 model_breathing: bool,
                 transposed_attn_convs: bool,
                 encoder_attention_conv_filters: list = None,
                 decoder_attention_conv_filters: list = None,
                 encoder_attention_conv_kernel: int = None,
                 decoder_attention_conv_kernel: int = None,
                 encoder_feed_forward_dimension: int = None,
                 decoder_feed_forward_dimension: int = None,
                 debug=False,
                 **kwargs):
        # reference model/models.py:345-440.  Unknown keys are tolerated and echoed in self.config.
        self.config = self._make_config(locals(), kwargs)
        self.device = torch.device(kwargs.get('device', 'cuda:0'))
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if self.device.type != 'cuda' or not torch.cuda.is_available():
            raise ops._lib.TtsmiError('ForwardTransformer runs on an MI355X through libttsmi.so; no GPU is '
                                      'visible and there is no CPU fallback')
        ops._lib.lib()                       # fail loudly now if the HIP library is missing
        self.text_pipeline = TextToTokens.default(phoneme_language, add_start_end=False,
                                                  with_stress=with_stress, model_breathing=model_breathing)
        self.symbols = self.text_pipeline.tokenizer.alphabet
        self.mel_channels = mel_channels
        self.vocab_size = self.text_pipeline.tokenizer.vocab_size
        c = self.config
        de, dd = encoder_model_dimension, decoder_model_dimension
        assert de == dd, 'Expand feeds the encoder width straight into the decoder (models.py:402,411)'
        spec = OrderedDict()
        spec['embedding'] = (self.vocab_size, de)
        spec.update(_blocks_spec('enc', de, encoder_num_heads, encoder_dense_blocks,
                                 encoder_feed_forward_dimension, encoder_attention_conv_filters,
                                 encoder_attention_conv_kernel))
        spec.update(_predictor_spec('dur', de, duration_conv_filters, duration_kernel_size))
        spec.update(_predictor_spec('pitch', de, pitch_conv_filters, pitch_kernel_size))
        spec['pitch_embed.w'] = (1, de)
        spec['pitch_embed.b'] = (de,)
        spec.update(_blocks_spec('dec', dd, decoder_num_heads, decoder_dense_blocks,
                                 decoder_feed_forward_dimension, decoder_attention_conv_filters,
                                 decoder_attention_conv_kernel))
        spec['out.w'] = (dd, mel_channels)
        spec['out.b'] = (mel_channels,)
        self.params = FlatParams(spec, self.device)
        self.pe_enc = torch.from_numpy(positional_encoding(encoder_max_position_encoding, de)).to(self.device)
        self.pe_dec = torch.from_numpy(positional_encoding(decoder_max_position_encoding, dd)).to(self.device)
        # optimiser state (reference: tf.keras Adam attached by _compile)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)     # optimizer.iterations
        self.lr_dev = torch.full((1,), 1e-4, dtype=torch.float32, device=self.device)
        self._host_step = 0
        self.beta_1, self.beta_2, self.epsilon = 0.9, 0.98, 1e-9
        self.loss_weights = [1., 1., 3.]
        self.loss = [masked_mean_absolute_error] * 3
        self.drop = ops.DropCtx(seed=int(kwargs.get('seed', 0)), step_dev=self.step_dev)
        # 'f32': exact-fp32 MFMA everywhere (the 1e-4 parity path).  'bf16': GEMM / attention operands
        # rounded to bf16, fp32 accumulate, fp32 master weights + activations (throughput path).
        self.precision = str(kwargs.get('precision', 'f32'))
        # 'bf16x3': the exact-fp32 path with its GEMM family on three bf16 MFMAs per product (ops.F32_GEMM_DTYPE); everything
        # else - attention, LayerNorm, residual stream, optimiser - is the fp32 path's
        assert self.precision in ('f32', 'bf16', 'bf16x3'), self.precision
        self.shadow: Dict[str, ops.Shadow] = {}
        self.overlap_wgrad = bool(kwargs.get('overlap_wgrad', True))   # wgrad on a second HIP stream
        if kwargs.get('use_graph', False):
            # (rounds 1-5 could replay the train step from captured hipGraphs: bit-identical, and slower than the eager step in
            # every round it was measured - 5.76 against 4.71 ms in round 5; with the step issued from C++ the host needs ~1.3 ms
            # for a 4.7 ms step and there is nothing left for a graph to hide.  predict() keeps its graphs: graph_inference.)
            raise ValueError('use_graph (hipGraph replay of train_step) was removed in round 6; the step is issued from C++ '
                             '(use_cstep, the default) instead')
        self.fused_blocks = bool(kwargs.get('fused_blocks', True))     # one autograd node per dense block
        self.overlap_predictors = bool(kwargs.get('overlap_predictors', True))   # StatPredictors on a side stream
        self._pred_stream, self._pred_pending, self._pred_keep, self._deferred_pred = None, False, None, None
        self._dropmask_plan, self._dropmask_bufs = {}, {}
        # blocks driven from C++ with persistent buffers (ops.DenseBlockPlan); only inside _forward_backward, where one
        # forward is followed by its backward before the next forward reuses the buffers
        # predict() replayed from hipGraphs (outputs are the graphs' static buffers: consume them before the next call)
        self.graph_inference = bool(kwargs.get('graph_inference', False))
        self._infer_graphs: Dict[tuple, dict] = {}
        self._const_masks: Dict[tuple, torch.Tensor] = {}
        self.planned_blocks = bool(kwargs.get('planned_blocks', True))
        self.fuse_ln = bool(kwargs.get('fuse_ln', True))               # res-norms in the GEMM epilogues (d_model 256)
        # consecutive planned blocks: the upper block's last dgrad runs the lower block's res-norm-2 backward in its epilogue
        self.chain_ln = bool(kwargs.get('chain_ln', True)) and os.environ.get('TTSMI_LN_CHAIN', '1') != '0'
        # bf16 residual stream inside the planned dense blocks (ttsmi_dense_block.res16): the residual adds of the fused
        # GEMM + LayerNorm kernels read the bf16 tensors the GEMMs read anyway, fp32 copies exist only at the stack ends
        self.residual_bf16 = bool(kwargs.get('residual_bf16', True)) and os.environ.get('TTSMI_RES16', '1') != '0'
        # bf16 precision, blocks outside the planned path: bf16 qkv / context tensors around the attention kernels
        self._attn_io_bf16 = os.environ.get('TTSMI_ATTN_IO_BF16', '1') != '0'
        self._use_plans, self._plans, self._plan_shared, self._plans_grown = False, {}, {}, {}
        # the forward's row-local chain of every planned dense block (o-projection + res-norm 1 -> FFN -> res-norm 2 -> the next
        # block's qkv projection) as ONE launch (csrc/chain.hip, chain16.h) instead of four, from ops.CHAIN_MIN_ROWS rows on
        # (decoder-size batches: 4.71 against 4.94 ms per step on the final build of round 5); chain_blocks=False / TTSMI_DENSE_CHAIN=0: always
        # the four launches
        self.chain_blocks = bool(kwargs.get('chain_blocks', os.environ.get('TTSMI_DENSE_CHAIN', '1') != '0'))
        self._weights_version = 0
        self._block_cache: Dict[str, tuple] = {}
        self.return_attention = None         # None = per-method default (see module docstring)
        self.grad_sync = None                # set by transformertts_amd.dp.DataParallel
        self.loss_denominators = None        # (mel, duration, pitch) element counts of the GLOBAL batch, set per step by dp.DataParallel
        self._lenreg_hook = None             # DataParallel: starts the decoder-half all-reduce (ops.LenRegFn.backward)
        # reference_outputs=True: train_step returns the 12 attention maps like the reference's _train_step
        # (models.py:544-549) instead of leaving the dicts empty - see the module docstring
        self.reference_outputs = bool(kwargs.get('reference_outputs', False))
        # the maps of a TRAIN step are written into a ring of MAP_RING_DEPTH persistent buffer sets per (block, shape)
        # instead of 2.7 GB of fresh tensors per step (config 2): a step's maps stay valid until MAP_RING_DEPTH - 1 more
        # steps have run (the trainer reads them right after the step, train_tts.py:175-176).  call / val_step / predict
        # keep returning fresh tensors.  map_ring=False restores fresh tensors everywhere.
        self.map_ring = bool(kwargs.get('map_ring', True)) and os.environ.get('TTSMI_MAP_RING', '1') != '0'
        self._map_ring_on, self._map_bufs = False, {}
        self.debug = debug
        # the train step as ONE descriptor issued from C++ (transformertts_amd/step.py; use_cstep=False / TTSMI_CSTEP=0: the
        # per-layer autograd path, which stays the path of everything the descriptor does not cover)
        self.use_cstep = bool(kwargs.get('use_cstep', True))
        self._cstep, self._cstep_eligible = None, None
        self._phase_events = None            # measurement instrumentation (_mark)
        self._taps = None                    # test instrumentation: a list receives (f'{prefix}.blk{i}', block output)
        self._init_weights(int(kwargs.get('seed', 0)))
        self._build_shadows()

    def _shadow_names(self):
        for name, w in self.params.w.items():
            leaf = name.split('.')[-1]
            if leaf in ('wqkv', 'wo', 'w1', 'w2') or name == 'out.w':
                yield name
            elif leaf == 'w' and w.dim() == 3:          # Conv1D weights: predictors and conv blocks
                yield name

    @_on_device
    def _build_shadows(self):
        """bf16 operand copies of the GEMM weights (precision == 'bf16' only)."""
        self.shadow = {}
        self.shadow_set = None
        if self.precision == 'bf16':
            P = self.params
            self.shadow_set = ops.ShadowSet(P.data, P.offsets, P.w, list(self._shadow_names()))
            self.shadow = self.shadow_set.sh
            self.shadow_set.refresh(False)

    def _refresh_shadows(self, wb_is_current: bool = False):
        if self.shadow_set is not None:
            self.shadow_set.refresh(wb_is_current)
        self._weights_version += 1                       # the chain kernels' weight streams are repacked on their next use

    # ------------------------------------------------------------------ construction helpers
    def _make_config(self, locals_: dict, kwargs: dict) -> dict:
        config = {}
        for k, v in locals_.items():
            if k in kwargs or k in ('self', '__class__', 'kwargs'):
                continue
            if isinstance(v, dict):
                config.update(v)
            else:
                config[k] = v
        config.update(kwargs)
        return config

    def _init_weights(self, seed: int):
        """Keras defaults [3P]: glorot_uniform Dense/Conv kernels (per original matrix: wq, wk, wv
        each [d,d]), zero biases, Embedding U(-0.05, 0.05), LN gamma=1 beta=0, pos scalar 1."""
        g = torch.Generator(device='cpu').manual_seed(seed)
        with torch.no_grad():
            for name, w in self.params.w.items():
                leaf = name.split('.')[-1]
                shape = tuple(w.shape)
                if name == 'embedding':
                    a = torch.rand(shape, generator=g) * 0.1 - 0.05
                elif leaf in ('gamma', 'pos_scalar'):
                    a = torch.ones(shape)
                elif leaf == 'beta' or (leaf.startswith('b') and len(shape) == 1):
                    a = torch.zeros(shape)
                elif leaf == 'wqkv':
                    d = shape[0]
                    lim = math.sqrt(6.0 / (2 * d))
                    a = (torch.rand(shape, generator=g) * 2 - 1) * lim
                elif len(shape) == 2:
                    lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                    a = (torch.rand(shape, generator=g) * 2 - 1) * lim
                elif len(shape) == 3:
                    k, cin, cout = shape
                    lim = math.sqrt(6.0 / (k * cin + k * cout))
                    a = (torch.rand(shape, generator=g) * 2 - 1) * lim
                else:
                    raise ValueError(name)
                w.copy_(a.to(self.device))

    # ------------------------------------------------------------------ weights interchange
    @_on_device
    def load_weights_dict(self, weights: Dict[str, np.ndarray]):
        """Load reference-named variables (separate wq/wk/wv, see oracle/ft_oracle.py:weight_spec -
        the Keras variable layout: Dense [in,out], Conv1D [k,in,out])."""
        with torch.no_grad():
            for name, w in self.params.w.items():
                leaf = name.split('.')[-1]
                if leaf == 'wqkv':
                    base = name[:-len('wqkv')]
                    a = np.concatenate([weights[base + 'wq'], weights[base + 'wk'], weights[base + 'wv']], 1)
                elif leaf == 'bqkv':
                    base = name[:-len('bqkv')]
                    a = np.concatenate([weights[base + 'bq'], weights[base + 'bk'], weights[base + 'bv']], 0)
                else:
                    a = weights[name]
                w.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(w.shape).to(self.device))
        self._refresh_shadows()

    def _export(self, tensors: Dict[str, torch.Tensor]) -> "OrderedDict[str, np.ndarray]":
        """Internal (fused wqkv/bqkv) -> reference variable names and order (wq,bq,wk,bk,wv,bv)."""
        out = OrderedDict()
        for name, t in tensors.items():
            leaf = name.split('.')[-1]
            if leaf == 'bqkv':
                continue
            a = t.detach().cpu().numpy().copy()
            if leaf == 'wqkv':
                base, d = name[:-len('wqkv')], a.shape[0]
                bias = tensors[base + 'bqkv'].detach().cpu().numpy().copy()
                for j, n in enumerate('qkv'):
                    out[base + 'w' + n] = a[:, j * d:(j + 1) * d]
                    out[base + 'b' + n] = bias[j * d:(j + 1) * d]
            else:
                out[name] = a
        return out

    def weights_dict(self) -> "OrderedDict[str, np.ndarray]":
        return self._export(self.params.w)

    def grads_dict(self) -> "OrderedDict[str, np.ndarray]":
        return self._export(self.params.g)

    # ------------------------------------------------------------------ layers (reference model/layers.py)
    def _self_attention_blocks(self, prefix, name, x, pad, klen, heads, dense_blocks, pe, rate,
                               want_attn):
        """SelfAttentionBlocks.call (layers.py:297-310) on fused kernels.  x [B,T,d]."""
        W, G, drop, S = self.params.w, self.params.g, self.drop, self.shadow.get
        B, T, d = x.shape
        M = B * T
        if T > pe.shape[0]:
            # the kernel indexes pe[(row % T) * d]; the reference fails on pos_encoding[:, :seq_len] (layers.py:300)
            raise ValueError(f'{name}: sequence length {T} exceeds the positional-encoding table '
                             f'({pe.shape[0]} positions; {prefix}oder_max_position_encoding)')
        # (planned dense blocks read the bf16 copy of their input: the stack's first LayerNorm writes it as well)
        first_planned = bool(heads) and dense_blocks > 0 and self.fused_blocks and self._use_plans and \
            self._plan_ok(f'{prefix}.blk0', heads[0], d)
        h = ops.add_layernorm(x.reshape(M, d), None, W[f'{prefix}.ln.gamma'], W[f'{prefix}.ln.beta'],
                              G[f'{prefix}.ln.gamma'], G[f'{prefix}.ln.beta'], pe=pe,
                              pe_scale=W[f'{prefix}.pos_scalar'], gpe_scale=G[f'{prefix}.pos_scalar'], T=T,
                              p_out=rate, site_out=drop.site(), drop=drop, want_h=first_planned)
        h, h_bf = h if first_planned else (h, None)     # bf16 copy of h: later written by the previous block's LayerNorm
        attn = OrderedDict()
        dtype = ops._lib.TTSMI_BF16 if self.precision == 'bf16' else ops.TTSMI_F32
        below = None                     # the planned block whose output feeds the current one (backward chaining)
        # consecutive planned blocks are issued by ONE C++ call per direction (ops.PlannedDenseStackFn) unless somebody
        # needs a block's output on the host side of the boundary (attention maps, activation taps)
        stack_mode = _DENSE_STACK and not want_attn and self._taps is None
        pending = []
        waited_ev = None                 # the keep-bit tables of a stack share ONE event: the main stream waits for it once

        def flush(h, h_bf):
            if not pending:
                return h, h_bf
            plans = tuple(pending)
            pending.clear()
            if len(plans) == 1:
                return ops.PlannedDenseBlockFn.apply(h, h_bf, plans[0])
            return ops.PlannedDenseStackFn.apply(h, h_bf, plans)
        for i, H in enumerate(heads):
            p = f'{prefix}.blk{i}'
            dense = i < dense_blocks
            if not (dense and self.fused_blocks and self._use_plans and self._plan_ok(p, H, d)):
                h, h_bf = flush(h, h_bf)
            if dense and self.fused_blocks and self._use_plans and self._plan_ok(p, H, d):
                # launch sequence of the block issued from C++ (ops.DenseBlockPlan): two host calls per block and step
                plan = self._block_plan(p, prefix, B, H, T)
                if below is not None:    # this block is the only consumer of the lower block's output (taps are detached)
                    below.chain_above(plan if self.chain_ln and torch.is_grad_enabled() else None)
                below = plan
                sites = (drop.site(), drop.site(), drop.site())
                dmask = None
                pre = self._dropmask_plan.get(p) if self._dropmask_plan else None
                if pre is not None:
                    dmask, site_planned, ev = pre
                    assert site_planned == sites[0], (p, site_planned, sites)
                    if ev is not None and ev is not waited_ev:
                        ops.cur_stream().wait_event(ev)
                        waited_ev = ev
                if h_bf is None:
                    h_bf = ops.to_bf16(h)
                nxt = i + 1
                last = not (nxt < len(heads) and nxt < dense_blocks and self._plan_ok(f'{prefix}.blk{nxt}', heads[nxt], d))
                # forward chaining: this block's chain launch also runs the next planned block's qkv projection
                plan.chain_forward(None if last else self._block_plan(f'{prefix}.blk{nxt}', prefix, B, heads[nxt], T))
                plan.bind(pad, klen, rate, drop, sites, dmask, res16=self.residual_bf16,
                          out32=last or self._taps is not None)
                plan.ensure_packed(self._weights_version)
                pending.append(plan)
                if not stack_mode or last:
                    h, h_bf = flush(h, h_bf)
                if want_attn:
                    attn[f'{name}_DenseBlock{i + 1}_SelfAttention'] = ops.attention_weights(
                        plan.t['qkv'], pad, plan.t['lse'], B, H, T, d // H, rate, drop, sites[0], ops._lib.TTSMI_BF16_IO,
                        dmask, out=self._map_buffer(p, B, H, T))
                if self._taps is not None:
                    self._taps.append((p, h.detach().reshape(B, T, d)))
                continue
            if below is not None:
                below.chain_above(None)
                below = None
            if dense and self.fused_blocks:
                # one autograd node per block (ops.DenseBlockFn); sites in the per-layer order
                Pb, Gb, Sb = self._block_views(p)
                sites = (drop.site(), drop.site(), drop.site())
                dmask = None
                pre = self._dropmask_plan.get(p) if self._dropmask_plan else None
                if pre is not None:                       # generated ahead of time on the side stream
                    dmask, site_planned, ev = pre
                    assert site_planned == sites[0], (p, site_planned, sites)
                    if ev is not None:
                        ops.cur_stream().wait_event(ev)
                h, h_bf, qkv, lse = ops.DenseBlockFn.apply(h, h_bf, Pb, Gb, Sb, pad, klen, B, H, T, rate, drop, sites,
                                                           dtype, want_attn, dmask)
                if h_bf.numel() == 0:
                    h_bf = None
                if want_attn:
                    attn[f'{name}_DenseBlock{i + 1}_SelfAttention'] = ops.attention_weights(
                        qkv, pad, lse, B, H, T, d // H, rate, drop, sites[0],
                        ops._lib.TTSMI_BF16_IO if qkv.dtype == torch.bfloat16 else ops.TTSMI_F32,
                        out=self._map_buffer(p, B, H, T))
                if self._taps is not None:
                    self._taps.append((p, h.detach().reshape(B, T, d)))
                continue
            # bf16 precision, per-layer path (the conv blocks of the reference-default architecture): qkv / ctx / their
            # gradients are bf16 tensors as in the planned dense blocks - the attention kernels on fp32 tensors (dh 192:
            # 99 KB of LDS, scratch) run the backward in 1.43 ms per decoder layer against 0.52 ms on bf16 tensors
            io_h = (self.precision == 'bf16' and self._attn_io_bf16 and (d // H) in (32, 64, 192) and d % 8 == 0
                    and S(f'{p}.wqkv') is not None and S(f'{p}.wo') is not None)
            if io_h and h_bf is None:
                h_bf = ops.to_bf16(h)
            # the block input's three gradients (qkv projection, q_in half of the output projection, residual) and the conv
            # stack input's two are summed in place (ops.GradSink) instead of by autograd's add launches
            sink_h = ops.GradSink(f'{p}.in').watch(h) if (_GRAD_SINK and h.requires_grad) else None
            sink_a = ops.GradSink(f'{p}.a') if (_GRAD_SINK and h.requires_grad) else None
            qkv = ops.LinearFn.apply(h, None, W[f'{p}.wqkv'], W[f'{p}.bqkv'], G[f'{p}.wqkv'], G[f'{p}.bqkv'],
                                     S(f'{p}.wqkv'), io_h, h_bf if io_h else None, sink_h)
            site = drop.site()
            dmask = None
            pre = self._dropmask_plan.get(p) if self._dropmask_plan else None
            if pre is not None:                  # keep-bit table generated ahead on the side stream (_launch_dropmasks)
                dmask, site_planned, ev = pre
                assert site_planned == site, (p, site_planned, site)
                if ev is not None:
                    ops.cur_stream().wait_event(ev)
            ctx, lse = ops.AttentionFn.apply(qkv, pad, klen, B, H, T, d // H, rate, drop, site,
                                             ops._lib.TTSMI_BF16 if self.precision == 'bf16' else ops.TTSMI_F32, dmask)
            if want_attn:
                key = (f'{name}_DenseBlock{i + 1}_SelfAttention' if dense
                       else f'{name}_ConvBlock{i - dense_blocks + 1}_SelfAttention')
                attn[key] = ops.attention_weights(qkv.detach(), pad, lse, B, H, T, d // H, rate, drop, site,
                                                  ops._lib.TTSMI_BF16_IO if io_h else ops.TTSMI_F32, dmask if io_h else None,
                                                  out=self._map_buffer(p, B, H, T))
            o = ops.LinearFn.apply(h, ctx, W[f'{p}.wo'], W[f'{p}.bo'], G[f'{p}.wo'], G[f'{p}.bo'], S(f'{p}.wo'),
                                   False, h_bf if io_h else None, sink_h)
            h_bf = None
            a = ops.add_layernorm(o, h, W[f'{p}.ln1.gamma'], W[f'{p}.ln1.beta'], G[f'{p}.ln1.gamma'],
                                  G[f'{p}.ln1.beta'], row_pad=pad, p_in=rate, site_in=drop.site(), drop=drop,
                                  res_sink=sink_h)
            if dense:
                f = ops.FFNFn.apply(a, W[f'{p}.ffn.w1'], W[f'{p}.ffn.b1'], W[f'{p}.ffn.w2'], W[f'{p}.ffn.b2'],
                                    G[f'{p}.ffn.w1'], G[f'{p}.ffn.b1'], G[f'{p}.ffn.w2'], G[f'{p}.ffn.b2'],
                                    S(f'{p}.ffn.w1'), S(f'{p}.ffn.w2'))
            else:
                n = 0
                while f'{p}.conv{n}.w' in W:
                    n += 1
                ps, gs = [], []
                for j in range(n):
                    ps += [W[f'{p}.conv{j}.w'], W[f'{p}.conv{j}.b']]
                    gs += [G[f'{p}.conv{j}.w'], G[f'{p}.conv{j}.b']]
                shs = tuple(S(f'{p}.conv{j}.w') for j in range(n))
                if sink_a is not None:
                    sink_a.watch(a)
                f = ops.ConvStackFn.apply(a.reshape(B, T, d), n, shs, *ps, *gs, sink_a).reshape(M, d)
            h = ops.add_layernorm(f, a, W[f'{p}.ln2.gamma'], W[f'{p}.ln2.beta'], G[f'{p}.ln2.gamma'],
                                  G[f'{p}.ln2.beta'], row_pad=pad, p_in=rate, site_in=drop.site(), drop=drop, want_h=io_h,
                                  res_sink=None if dense else sink_a)
            if io_h:
                h, h_bf = h             # the next block's bf16 GEMM operand, written by the same LayerNorm launch
            if self._taps is not None:
                self._taps.append((p, h.detach().reshape(B, T, d)))
        h, h_bf = flush(h, h_bf)
        if below is not None:
            below.chain_above(None)          # the stack's last block: its output gradient comes from outside
        return h.reshape(B, T, d), attn

    def _plan_ok(self, p, H, d) -> bool:
        return (self.precision == 'bf16' and (d // H) in (32, 64) and d % 64 == 0
                and all(f'{p}.{k}' in self.shadow for k in ('wqkv', 'wo', 'ffn.w1', 'ffn.w2')))

    def _block_plan(self, p, prefix, B, H, T):
        """ops.DenseBlockPlan of block `p`, re-targeted at this batch shape.  One plan per block and mode (training /
        forward-only) sized for the largest row count seen so far: a new (B, T) only re-binds it.  When a batch needs
        more rows than the capacity the whole stack's plans are rebuilt (after a device synchronisation: nothing in
        flight reads the dropped buffers; captured graphs keep the plans they baked in alive themselves)."""
        backward = torch.is_grad_enabled()
        key = (p, 'bwd' if backward else 'fwd')
        plan = self._plans.get(key)
        M = B * T
        if plan is not None and plan.cap < M:
            torch.cuda.synchronize()
            for k in [k for k in self._plans if k[0].startswith(prefix + '.') and k[1] == key[1]]:
                del self._plans[k]
            self._plan_shared.get((prefix, key[1]), {}).clear()
            plan = None
            # graphs that baked the dropped plans' buffers in are dropped with them and re-captured on their next use
            # (they used to pin a complete copy of the stack's buffers per growth event: HBM grew with the number of
            # growth events under graph mode - advisor finding, round 3)
            if key[1] != 'fwd':
                pass
            elif prefix == 'enc':
                self._infer_graphs.clear()               # graph A of every input shape holds the encoder plans
            else:
                for A in self._infer_graphs.values():    # graphs B (one per decoder-length bucket) hold the decoder plans
                    A['B'].clear()
        if plan is None:
            Pb, Gb, Sb = self._block_views(p)
            # every block of a stack gets the same capacity (the first one to be built decides; a little headroom so
            # that a slowly growing maximum does not rebuild the stack every few steps)
            cap = max([pl.cap for k, pl in self._plans.items() if k[0].startswith(prefix + '.') and k[1] == key[1]] +
                      [M if not self._plans_grown.get((prefix, key[1])) else (M * 3 + 1) // 2])     # geometric growth: few rebuilds
            self._plans_grown[(prefix, key[1])] = True
            plan = self._plans[key] = ops.DenseBlockPlan(Pb, Gb, Sb, B, H, T, self.device,
                                                         self._plan_shared.setdefault((prefix, key[1]), {}), self.fuse_ln,
                                                         backward=backward, cap_rows=cap,
                                                         chain=self.chain_blocks and self.residual_bf16)
        else:
            plan.rebind(B, T)
        return plan

    _BLOCK_KEYS = ('wqkv', 'bqkv', 'wo', 'bo', 'ln1.gamma', 'ln1.beta', 'ffn.w1', 'ffn.b1', 'ffn.w2', 'ffn.b2',
                   'ln2.gamma', 'ln2.beta')

    def _block_views(self, p):
        """(parameters, gradient sinks, bf16 shadows) of one dense block, keyed by their local names."""
        v = self._block_cache.get(p)
        if v is None:
            W, G = self.params.w, self.params.g
            v = ({k: W[f'{p}.{k}'] for k in self._BLOCK_KEYS}, {k: G[f'{p}.{k}'] for k in self._BLOCK_KEYS},
                 {k: self.shadow[f'{p}.{k}'] for k in self._BLOCK_KEYS if f'{p}.{k}' in self.shadow})
            self._block_cache[p] = v
        return v

    def _stat_predictor(self, prefix, x, pad, n_layers, relu_head, rate):
        """StatPredictor.call + CNNDropout.call (layers.py:481-485,510-524).  x [B,T,d]."""
        W, G, drop = self.params.w, self.params.g, self.drop
        if _PRED_ONE_NODE:
            convs = [(W[f'{prefix}.conv{j}.w'], W[f'{prefix}.conv{j}.b'], G[f'{prefix}.conv{j}.w'], G[f'{prefix}.conv{j}.b'],
                      self.shadow.get(f'{prefix}.conv{j}.w')) for j in range(n_layers)]
            lns = [(W[f'{prefix}.ln{j}.gamma'], W[f'{prefix}.ln{j}.beta'], G[f'{prefix}.ln{j}.gamma'], G[f'{prefix}.ln{j}.beta'],
                    drop.site()) for j in range(n_layers)]
            lin = (W[f'{prefix}.lin.w'], W[f'{prefix}.lin.b'], G[f'{prefix}.lin.w'], G[f'{prefix}.lin.b'])
            return ops.StatPredictorFn.apply(x, pad, convs, lns, lin, relu_head, rate, drop)
        B, T, _ = x.shape
        h = ops.RowMaskFn.apply(x, pad)
        for j in range(n_layers):
            h = ops.ConvReluPreMaskedFn.apply(h, W[f'{prefix}.conv{j}.w'], W[f'{prefix}.conv{j}.b'],
                                              G[f'{prefix}.conv{j}.w'], G[f'{prefix}.conv{j}.b'],
                                              self.shadow.get(f'{prefix}.conv{j}.w'))
            C = h.shape[-1]
            h = ops.add_layernorm(h.reshape(B * T, C), None, W[f'{prefix}.ln{j}.gamma'],
                                  W[f'{prefix}.ln{j}.beta'], G[f'{prefix}.ln{j}.gamma'],
                                  G[f'{prefix}.ln{j}.beta'], p_out=rate, site_out=drop.site(), drop=drop,
                                  relu_in=True).reshape(B, T, C)
        return ops.RowDotFn.apply(h, W[f'{prefix}.lin.w'], W[f'{prefix}.lin.b'], G[f'{prefix}.lin.w'],
                                  G[f'{prefix}.lin.b'], pad, relu_head)

    # ------------------------------------------------------------------ reference model/models.py:518-550
    @_on_device
    def call(self, x, target_durations=None, target_pitch=None, training=False, durations_scalar=1.,
             max_durations_mask=None, min_durations_mask=None, mel_len: Optional[int] = None,
             return_attention: Optional[bool] = None, _overlap_predictors: bool = False):
        want_attn = (not training) if return_attention is None else return_attention
        a = self._call_front(x, target_durations, target_pitch, training, durations_scalar, max_durations_mask,
                             min_durations_mask, want_attn, _overlap_predictors, need_total=mel_len is None)
        if mel_len is None:
            # inference: the output length max_b sum(round(dur)) is data dependent (one host sync,
            # like the reference's eager call)
            mel_len = max(int(a['total'].max().item()), 1)
        b = self._call_back(a['h'], a['use'], mel_len, training, want_attn)
        if self._deferred_pred is not None:          # teacher-forced training: the predictors, issued after the decoder
            a['duration'], a['pitch'] = self._deferred_pred()
            self._deferred_pred = None
        return {'mel': b['mel'], 'duration': a['duration'], 'pitch': a['pitch'], 'expanded_mask': b['expanded_mask'],
                'encoder_attention': a['encoder_attention'], 'decoder_attention': b['decoder_attention'],
                'expanded_lengths': b['expanded_lengths']}

    def _call_front(self, x, target_durations, target_pitch, training, durations_scalar, max_durations_mask,
                    min_durations_mask, want_attn, _overlap_predictors=False, need_total=True):
        """models.py:521-539: masks, embedding, encoder, predictors, pitch embedding, the durations to expand by."""
        c, W, G = self.config, self.params.w, self.params.g
        rate = c['dropout_rate'] if training else 0.0
        prate = c['predictors_dropout'] if training else 0.0
        self.drop.reset()
        x = torch.as_tensor(x, device=self.device).to(torch.int32).contiguous()
        B, Tp = x.shape
        pad_e, klen_e = ops.token_pad_mask(x)                                        # :521
        h = ops.EmbeddingFn.apply(x, W['embedding'], G['embedding'])                 # :522
        self._mark('start')
        h, enc_attn = self._self_attention_blocks('enc', 'Encoder', h, pad_e, klen_e,
                                                  c['encoder_num_heads'], c['encoder_dense_blocks'],
                                                  self.pe_enc, rate, want_attn)      # :523
        self._mark('enc_fwd')
        # With teacher forcing (training / validation: target durations AND target pitch given) nothing downstream of
        # the two StatPredictors feeds the decoder - their outputs only meet the losses.  They are ~60 small launches
        # (M = B*Tp rows) that cannot fill the GPU, so they go to a second HIP stream and run underneath the decoder;
        # autograd replays each node's backward on the stream its forward ran on, so their backward overlaps too.
        # (only from _forward_backward, which owns the joins: before the losses, and again after backward)
        overlap_pred = (_overlap_predictors and self.overlap_predictors and target_durations is not None
                        and target_pitch is not None)
        self._deferred_pred = None
        if overlap_pred:
            main = ops.cur_stream()
            if self._pred_stream is None:
                self._pred_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('TTSMI_PRED_PRIO', '0')))
            side = self._pred_stream
            side.wait_stream(main)                       # the encoder output is complete in main-stream order
            self._pred_keep = [h, pad_e]                 # the side stream reads them: alive until the join
            n_dur, n_pit = len(c['duration_conv_filters']), len(c['pitch_conv_filters'])

            def run_predictors(h=h, pad_e=pad_e):
                with ops.on_stream(side):
                    hb = ops.BranchFn.apply(h) if h.requires_grad else h      # one gradient edge, summed on the side stream
                    d_ = self._stat_predictor('dur', hb, pad_e, n_dur, True, prate)
                    p_ = self._stat_predictor('pitch', hb, pad_e, n_pit, False, prate)
                self._pred_pending = True
                return d_, p_
            if _PRED_LATE:
                # ISSUED after the decoder (call() runs the closure once _call_back has enqueued it): the autograd engine
                # replays nodes newest-first, so the predictors' backward - ~60 small launches on the side stream - is then
                # issued at the START of the backward pass and hides under the decoder's, instead of starting when the
                # decoder's backward is nearly done and leaving the main stream ~0.25 ms idle at the encoder boundary,
                # where d(encoder output) needs all three of its consumers (kernel trace).  The
                # dropout sites keep their numbers: they are reserved here and restored for the deferred call.
                site0 = self.drop._site
                self.drop._site += n_dur + n_pit

                def deferred():
                    keep = self.drop._site
                    self.drop._site = site0
                    try:
                        return run_predictors()
                    finally:
                        self.drop._site = keep
                self._deferred_pred = deferred
                durations = pitch = None
            else:
                durations, pitch = run_predictors()
        else:
            durations = self._stat_predictor('dur', h, pad_e, len(c['duration_conv_filters']), True, prate)
            pitch = self._stat_predictor('pitch', h, pad_e, len(c['pitch_conv_filters']), False, prate)
        if target_pitch is not None:                                                 # :527-530
            p_in = torch.as_tensor(target_pitch, device=self.device).to(torch.float32).reshape(B * Tp)
        else:
            p_in = pitch.reshape(B * Tp)
        h = ops.PitchEmbedFn.apply(h, p_in, W['pitch_embed.w'].reshape(-1), W['pitch_embed.b'],
                                   G['pitch_embed.w'].reshape(-1), G['pitch_embed.b'])   # :531
        if target_durations is not None:                                             # :532-535
            use = torch.as_tensor(target_durations, device=self.device).reshape(B, Tp)
            if use.dtype not in (torch.int32, torch.float32):
                use = use.to(torch.int32 if not use.dtype.is_floating_point else torch.float32)
        else:
            use = (durations.detach() * float(durations_scalar)).reshape(B, Tp)
        if max_durations_mask is not None:                                           # :536-537
            use = torch.minimum(use.to(torch.float32),
                                torch.as_tensor(max_durations_mask, device=self.device).to(torch.float32))
        if min_durations_mask is not None:                                           # :538-539
            use = torch.maximum(use.to(torch.float32),
                                torch.as_tensor(min_durations_mask, device=self.device).to(torch.float32))
        total = ops.lenreg_index(use, 1)[2] if need_total else None      # [B] sum(round(dur)): the output length
        return {'h': h, 'use': use, 'total': total, 'duration': durations, 'pitch': pitch, 'encoder_attention': enc_attn}

    def _call_back(self, h, use, mel_len, training, want_attn):
        """models.py:540-543: Expand, decoder mask, decoder, mel projection (dropout sites continue _call_front's)."""
        c, W, G = self.config, self.params.w, self.params.g
        rate = c['dropout_rate'] if training else 0.0
        B = h.shape[0]
        idx, cum, lens = ops.lenreg_index(use, mel_len)                              # :540 Expand
        mels = ops.LenRegFn.apply(h, idx, cum, self._lenreg_hook)
        pad_d, klen_d = ops.length_pad_mask(lens, mel_len)                           # :541
        expanded_mask = pad_d.to(torch.float32)[:, None, None, :]
        mels, dec_attn = self._self_attention_blocks('dec', 'Decoder', mels, pad_d, klen_d,
                                                     c['decoder_num_heads'], c['decoder_dense_blocks'],
                                                     self.pe_dec, rate, want_attn)   # :542
        self._mark('dec_fwd')
        out = ops.LinearFn.apply(mels.reshape(B * mel_len, -1), None, W['out.w'], W['out.b'], G['out.w'],
                                 G['out.b'], self.shadow.get('out.w')).reshape(B, mel_len, self.mel_channels)  # :543
        return {'mel': out, 'expanded_mask': expanded_mask, 'decoder_attention': dec_attn, 'expanded_lengths': lens}

    __call__ = call

    def _launch_dropmasks(self, B, Tp, Tm, rate):
        """Attention-dropout keep bits of every dense block of this step, generated on the side stream ahead of their
        use: the generator is pure VALU work (the same hash the kernels would otherwise evaluate in their inner loops,
        ~35 us per decoder layer) and hides under the encoder's small launches.  Two events: encoder tables first."""
        self._dropmask_plan = {}
        c = self.config
        if not (self.precision == 'bf16' and rate > 0 and ops._ATTN_DROPBITS and self.fused_blocks):
            return
        d = c['encoder_model_dimension']
        l = ops._lib.lib()
        main = ops.cur_stream()
        if self._pred_stream is None:
            self._pred_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('TTSMI_PRED_PRIO', '0')))
        side = self._pred_stream
        side.wait_stream(main)                 # the step counter was advanced, and last step's readers are done
        # dropout-site numbering of call(): one site for each stack's entry LayerNorm, three per block (attention,
        # ln1, ln2), one per predictor layer between the two stacks
        site = 1
        plans = []
        for prefix, heads, nd, T in (('enc', c['encoder_num_heads'], c['encoder_dense_blocks'], Tp),
                                     ('dec', c['decoder_num_heads'], c['decoder_dense_blocks'], Tm)):
            if prefix == 'dec':
                site += len(c['duration_conv_filters']) + len(c['pitch_conv_filters']) + 1
            for i, H in enumerate(heads):
                dh = d // H
                # dense blocks only: for the conv blocks' attention sub-layer (fp32 tensors, dh = 192: three dK/dV passes)
                # the table was measured and does NOT pay - 31.8 vs 31.1 ms per ref-default step - although
                # ops.AttentionFn accepts one (TTSMI_ATTN_DROPBITS_CONV=1 turns it on)
                if (i < nd and dh in (32, 64) and d % 64 == 0) or (i >= nd and dh in (32, 64, 192) and _DROPBITS_CONV):
                    plans.append((f'{prefix}.blk{i}', H, T, site + 1))
                site += 3
        with ops.on_stream(side):
            for prefix in ('enc', 'dec'):          # encoder tables first, with their own event: the encoder starts
                ev = None                          # ~50 us into the step, the decoder ~1 ms
                for name, H, T, st in plans:
                    if not name.startswith(prefix):
                        continue
                    # one table per block, sized for the largest (B, H, T) seen: a smaller shape uses a prefix (length-
                    # bucketed data brings a new shape almost every step - a table per exact shape grew without bound)
                    need = int(l.ttsmi_attention_dropmask_bytes(B, H, T))
                    buf = self._dropmask_bufs.get(name)
                    if buf is None or buf.numel() < need:
                        # (the old table may still be read by the previous step: it is freed in stream order on the side
                        # stream that wrote it, and captured graphs pin their own, see _train_step_graphed)
                        buf = self._dropmask_bufs[name] = torch.empty(need, dtype=torch.uint8, device=self.device)
                    ops.attention_dropmask(B, H, T, rate, self.drop, st, self.device, out=buf)
                    if ev is None:
                        ev = torch.cuda.Event()
                    self._dropmask_plan[name] = (buf, st, ev)
                if ev is not None:
                    ev.record(side)

    def _launch_chain_packs(self):
        """The chain kernels' weight streams (ops.DenseBlockPlan.ensure_packed: 12 launches of ~3 us per step, one after the
        other in front of the decoder stack) repacked on the side stream instead, under the encoder's small launches: the
        plans that ran their chains in the previous step are packed for this step's weights now, and the main stream waits
        for one event when the first of them is reached.  A plan that is new, or whose shape falls under the row threshold
        this time, takes the in-line path as before."""
        if not (_CHAIN_PREPACK and self.chain_blocks and self.planned_blocks) or torch.cuda.is_current_stream_capturing():
            return
        todo = [pl for pl in self._plans.values() if pl.chain_on and pl.backward and pl.packed_ver != self._weights_version]
        if not todo:
            return
        main = ops.cur_stream()
        if self._pred_stream is None:
            self._pred_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get('TTSMI_PRED_PRIO', '0')))
        side = self._pred_stream
        side.wait_stream(main)                 # the refreshed bf16 shadows, and the previous step's readers of the old streams
        with ops.on_stream(side):
            for pl in todo:
                pl.ensure_packed(self._weights_version)
            ev = torch.cuda.Event()
            ev.record(side)
        for pl in todo:
            pl.pack_ev = ev

    MAP_RING_DEPTH = 2
    MAP_RING_SHAPES = 8        # (block, B, H, T) entries kept per block: ragged batches bring a new shape almost every step

    def _map_buffer(self, block, B, H, T):
        """The [B,H,T,T] fp32 buffer one attention map of this TRAIN step is written into (None: a fresh tensor)."""
        if not self._map_ring_on:
            return None
        slots = self._map_bufs.setdefault(block, OrderedDict())
        key = (B, H, T)
        ring = slots.get(key)
        if ring is None:
            if len(slots) >= self.MAP_RING_SHAPES:
                slots.popitem(last=False)                 # least recently used shape: freed in stream order
            ring = slots[key] = [None] * self.MAP_RING_DEPTH
        else:
            slots.move_to_end(key)
        i = self._host_step % self.MAP_RING_DEPTH
        if ring[i] is None:
            ring[i] = torch.empty((B, H, T, T), dtype=torch.float32, device=self.device)
        return ring[i]

    def _mark(self, name):
        """Measurement hook (tools/probe_phases.py): a timing event on the main stream at a phase boundary."""
        if self._phase_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._phase_events.append((name, ev))

    def _join_predictors(self):
        """Main stream waits for the predictor side stream (no-op when nothing is in flight there)."""
        if self._pred_pending:
            ops.cur_stream().wait_stream(self._pred_stream)
        self._pred_keep = None

    # ------------------------------------------------------------------ steps (models.py:464-507)
    def _losses(self, model_out, target_sequence, target_durations, target_pitch, unit_seed=False):
        return weighted_sum_losses((target_sequence, target_durations, target_pitch),
                                   (model_out['mel'], model_out['duration'], model_out['pitch']),
                                   self.loss, self.loss_weights, unit_seed=unit_seed,
                                   denominators=self.loss_denominators)

    def _prep(self, input_sequence, target_sequence, target_durations, target_pitch):
        dev = self.device
        x = torch.as_tensor(input_sequence, device=dev).to(torch.int32).contiguous()
        ts = torch.as_tensor(target_sequence, device=dev).to(torch.float32).contiguous()
        td = torch.as_tensor(target_durations, device=dev).to(torch.int32)[..., None].contiguous()   # :465
        tp = torch.as_tensor(target_pitch, device=dev).to(torch.float32)[..., None].contiguous()     # :466
        return x, ts, td, tp

    @_on_device
    def _train_step(self, input_sequence, target_sequence, target_durations, target_pitch):
        """reference _train_step models.py:464-482: forward(training=True), weighted L1 losses,
        backward, one TF-form Adam step.  Requires sum_b(dur) <= mel_len (the reference data has
        sum(dur_b) == mel_len_b; frames past mel_len would be sliced away at models.py:473)."""
        x, ts, td, tp = self._prep(input_sequence, target_sequence, target_durations, target_pitch)
        if self._cstep_ok():
            return self._train_step_c(x, ts, td, tp)
        model_out = self._forward_backward(x, ts, td, tp)
        if self.grad_sync is not None:
            self.grad_sync(self.params.grad)          # the single RCCL all-reduce of the step
        self._apply_gradients()                                                      # :481
        self._host_step += 1
        return model_out

    def _cstep_ok(self) -> bool:
        """The step as one descriptor issued from C++ (transformertts_amd/step.py) covers this model and this call: bf16
        planned dense blocks throughout, no attention maps returned, no instrumentation that lives in the per-layer path."""
        if self._cstep_eligible is None:
            from .. import step as _step
            self._cstep_eligible = bool(self.use_cstep) and _step.eligible(self)
        ra = self.reference_outputs if self.return_attention is None else self.return_attention
        # (a C-ABI trace hook - bench.py's instrumented step - wants to see every entry point: the per-layer path shows them)
        return (self._cstep_eligible and not ra and self._taps is None and self._phase_events is None
                and ops._lib._trace is None and not torch.cuda.is_current_stream_capturing() and torch.is_grad_enabled())

    def _train_step_c(self, x, ts, td, tp):
        """_train_step on a TrainStepPlan: three C calls (forward + loss + decoder backward / the rest of the backward /
        optimiser), the data-parallel all-reduces between them where the per-layer path launches them."""
        from .. import step as _step
        if self._cstep is None:
            self._cstep = _step.TrainStepPlan(self)
        plan = self._cstep
        with ops.pinned_stream():
            if self.grad_sync is None and self._lenreg_hook is None:
                out = plan.run(x, ts, td, tp)
            else:
                out = plan.run(x, ts, td, tp, phases=(0,))
                if self._lenreg_hook is not None:
                    plan.flush_decoder_ln()      # the decoder half's LayerNorm gradients are final before their all-reduce
                    W = ops._WgradStream.cur() if self.overlap_wgrad else None
                    if W is not None:
                        W.pending = True         # (the hook orders its collective behind the weight-gradient stream too)
                    self._lenreg_hook()
                    if W is not None:
                        W.pending = False        # phase 1 joins that stream itself
                plan.run(x, ts, td, tp, phases=(1,))
                if self.grad_sync is not None:
                    self.grad_sync(self.params.grad)
                plan.run(x, ts, td, tp, phases=(2,))
        self._host_step += 1
        return out

    def _forward_backward(self, x, ts, td, tp):
        """forward(training=True) + losses + backward into the flat gradient buffer.  No host sync,
        no allocation outside torch's allocator: hipGraph-capturable as one unit."""
        mel_len = int(ts.shape[1])                                                   # :467
        ra = self.reference_outputs if self.return_attention is None else self.return_attention
        with ops.pinned_stream():
            self._launch_dropmasks(int(x.shape[0]), int(x.shape[1]), mel_len, float(self.config['dropout_rate']))
            self._launch_chain_packs()
            self._use_plans = self.planned_blocks
            self._map_ring_on = bool(ra) and self.map_ring and not torch.cuda.is_current_stream_capturing()
            try:
                model_out = self.call(x, td, target_pitch=tp, training=True, mel_len=mel_len, return_attention=ra,
                                      _overlap_predictors=True)
            finally:
                self._dropmask_plan = {}
                self._use_plans = False
                self._map_ring_on = False
            self._join_predictors()              # the duration / pitch losses read the side stream's outputs
            loss, loss_vals = self._losses(model_out, ts, td, tp, unit_seed=True)    # seeded by loss.backward() below
            ops.enable_wgrad_stream(self.overlap_wgrad)
            try:
                with ops.ln_param_batch():
                    loss.backward()                                                  # :480
                    self._mark('bwd')
                    ops.ln_flush()               # LayerNorm parameter gradients: one reduce per producing stream, then
                    self._join_predictors()      # the main stream waits for the predictor stream ...
                    self._pred_pending = False
                ops.wgrad_join()                 # ... and for the weight-gradient stream, before all-reduce and Adam
            finally:
                ops.enable_wgrad_stream(False)
                self._pred_pending = False
        model_out = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in model_out.items()}
        model_out.update({'loss': loss.detach()})
        model_out.update({'losses': {'mel': loss_vals[0].detach(), 'duration': loss_vals[1].detach(),
                                     'pitch': loss_vals[2].detach()}})
        return model_out

    @_on_device
    def _val_step(self, input_sequence, target_sequence, target_durations, target_pitch):
        x, ts, td, tp = self._prep(input_sequence, target_sequence, target_durations, target_pitch)
        mel_len = int(ts.shape[1])
        ra = True if self.return_attention is None else self.return_attention
        with torch.no_grad():
            model_out = self.call(x, td, target_pitch=tp, training=False, mel_len=mel_len, return_attention=ra)
            loss, loss_vals = self._losses(model_out, ts, td, tp)
        model_out.update({'loss': loss})
        model_out.update({'losses': {'mel': loss_vals[0], 'duration': loss_vals[1], 'pitch': loss_vals[2]}})
        return model_out

    train_step = _train_step
    val_step = _val_step

    @_on_device
    def _apply_gradients(self):
        """tf.keras Adam(lr, 0.9, 0.98, 1e-9) (utils/training_config_manager.py:102-106) as one fused
        launch over the flat buffers; iteration counter and lr live on the device."""
        ops.step_increment(self.step_dev)
        P = self.params
        ss = self.shadow_set
        ops.adam_tf(P.data, P.grad, P.m, P.v, self.lr_dev, self.step_dev, self.beta_1, self.beta_2,
                    self.epsilon, shadow=None if ss is None else ss.flat_bf16)
        self._refresh_shadows(wb_is_current=True)
        self._mark('adam')

    def _compile(self, optimizer=None, learning_rate: Optional[float] = None):
        """reference _compile models.py:484-490.  `optimizer` may be any object with
        learning_rate/lr, beta_1, beta_2, epsilon attributes (e.g. the Adam config the reference
        builds in training_config_manager.py:102-106); only TF-form Adam is implemented."""
        self.loss_weights = [1., 1., 3.]
        if optimizer is not None:
            for src, dst in (('beta_1', 'beta_1'), ('beta_2', 'beta_2'), ('epsilon', 'epsilon')):
                if hasattr(optimizer, src):
                    setattr(self, dst, float(getattr(optimizer, src)))
            lr = getattr(optimizer, 'learning_rate', getattr(optimizer, 'lr', None))
            if lr is not None:
                learning_rate = float(lr)
        if learning_rate is not None:
            self.set_constants(learning_rate=learning_rate)

    @property
    def step(self) -> int:                                                           # :514-516
        return self._host_step

    def set_constants(self, learning_rate: float = None, **kwargs):                  # :552-554
        if learning_rate is not None:
            self.lr_dev.fill_(float(learning_rate))

    def _forward(self, input_sequence, durations_scalar):                            # :509-512
        with torch.no_grad():
            return self.call(input_sequence, target_durations=None, target_pitch=None, training=False,
                             durations_scalar=durations_scalar)

    forward = _forward

    # ------------------------------------------------------------------ inference (models.py:556-595)
    def encode_text(self, text):
        return self.text_pipeline(text)

    @_on_device
    def predict(self, inp, encode=True, speed_regulator=1., phoneme_max_duration=None,
                phoneme_min_duration=None, max_durations_mask=None, min_durations_mask=None,
                phoneme_durations=None, phoneme_pitch=None):
        if encode:
            inp = self.encode_text(inp)
        if not (torch.is_tensor(inp) and inp.device == self.device):     # tokens already on the GPU stay there
            inp = torch.as_tensor(np.asarray(inp.cpu() if torch.is_tensor(inp) else inp), device=self.device)
        if inp.dim() < 2:
            inp = inp[None]
        inp = inp.to(torch.int32)
        duration_scalar = float(1. / speed_regulator)
        max_durations_mask = self._make_max_duration_mask(inp, phoneme_max_duration)
        min_durations_mask = self._make_min_duration_mask(inp, phoneme_min_duration)
        ra = True if self.return_attention is None else self.return_attention
        # forward-only use of the C++-driven blocks (persistent per-shape buffers: the outputs of the previous predict
        # of the same shape are overwritten)
        self._use_plans = self.planned_blocks and self.precision == 'bf16'
        try:
            if self.graph_inference:
                out = self._predict_graphed(inp, duration_scalar, max_durations_mask, min_durations_mask,
                                            phoneme_durations, phoneme_pitch, ra)
            else:
                with torch.no_grad():
                    out = self.call(inp, target_durations=phoneme_durations, target_pitch=phoneme_pitch,
                                    training=False, durations_scalar=duration_scalar,
                                    max_durations_mask=max_durations_mask, min_durations_mask=min_durations_mask,
                                    return_attention=ra)
        finally:
            self._use_plans = False
        out['mel'] = out['mel'].squeeze()                                            # :576
        return out

    MEL_BUCKET = 64        # graph_inference: decoder lengths are rounded up to a multiple of this many frames
    GRAPH_CACHE_SHAPES = 32   # graph_inference: input shapes whose graphs (and the buffers baked into them) are kept;
    #                           the least recently used shape is dropped beyond that, so a server fed arbitrary sentence
    #                           lengths holds a bounded amount of HBM

    def _predict_graphed(self, inp, duration_scalar, max_mask, min_mask, phoneme_durations, phoneme_pitch, ra):
        """predict() replayed from two captured hipGraphs (BASELINE.json configs[4]): the eager forward is ~250
        launches driven from Python - at batch 1 the host loop, not the GPU, is the latency.  The output length is data
        dependent (sum of the rounded durations), so the forward is cut where the reference's eager call synchronises
        anyway: graph A = masks .. encoder .. predictors .. pitch embedding .. total lengths (per input shape), one host
        read of the maximum length, graph B = Expand .. decoder .. mel projection per decoder-length BUCKET (multiples
        of MEL_BUCKET frames; rows past the true length are padding: masked keys, sliced off the outputs).
        The returned tensors are views of the graphs' static output buffers: they are valid until the next predict()
        of the same input shape (clone what must outlive it) - the price of a launch-free replay."""
        B, Tp = inp.shape
        dev = self.device
        f32 = lambda t: None if t is None else torch.as_tensor(t, device=dev).to(torch.float32).reshape(B, Tp).contiguous()
        durs = None
        if phoneme_durations is not None:
            durs = torch.as_tensor(phoneme_durations, device=dev).reshape(B, Tp)
            durs = durs.to(torch.float32 if durs.dtype.is_floating_point else torch.int32).contiguous()
        pit = f32(phoneme_pitch)
        keyA = ('A', B, Tp, float(duration_scalar), None if durs is None else durs.dtype, pit is not None, bool(ra))
        A = self._infer_graphs.get(keyA)
        if A is None:
            A = {'tok': inp.clone(), 'maxm': max_mask.clone(), 'minm': min_mask.clone(),
                 'dur': None if durs is None else durs.clone(), 'pit': None if pit is None else pit.clone(), 'B': {}}
            run = lambda: self._call_front(A['tok'], A['dur'], A['pit'], False, duration_scalar, A['maxm'], A['minm'], ra)
            with torch.no_grad():
                run()                                        # warm-up: lazy allocations happen outside the capture
                torch.cuda.synchronize()
                A['graph'] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(A['graph']):
                    A['out'] = run()
            A['plans'] = list(self._plans.values())          # the encoder plans' buffers are baked into graph A
            self._infer_graphs[keyA] = A
            while len(self._infer_graphs) > max(1, int(self.GRAPH_CACHE_SHAPES)):
                torch.cuda.synchronize()                     # no replay in flight still reads the dropped buffers
                self._infer_graphs.pop(next(iter(self._infer_graphs)))
        else:
            self._infer_graphs[keyA] = self._infer_graphs.pop(keyA)      # most recently used last
        for dst, src in ((A['tok'], inp), (A['maxm'], max_mask), (A['minm'], min_mask), (A['dur'], durs), (A['pit'], pit)):
            if dst is not None:
                dst.copy_(src, non_blocking=True)
        A['graph'].replay()
        a = A['out']
        L = max(int(a['total'].max().item()), 1)             # the one host sync of the call (as in the eager reference)
        cap = int(self.pe_dec.shape[0])
        if L > cap:
            raise ValueError(f'Decoder: sequence length {L} exceeds the positional-encoding table ({cap} positions)')
        bucket = min((L + self.MEL_BUCKET - 1) // self.MEL_BUCKET * self.MEL_BUCKET, cap)
        Bg = A['B'].get(bucket)
        if Bg is None:
            run = lambda: self._call_back(a['h'], a['use'], bucket, False, ra)
            with torch.no_grad():
                run()
                torch.cuda.synchronize()
                Bg = {'graph': torch.cuda.CUDAGraph()}
                with torch.cuda.graph(Bg['graph'], pool=A['graph'].pool()):
                    Bg['out'] = run()
            Bg['plans'] = list(self._plans.values())     # their buffers are baked into the graphs: alive as long as the graph
            A['B'][bucket] = Bg
        Bg['graph'].replay()
        b = Bg['out']
        out = {'mel': b['mel'][:, :L], 'duration': a['duration'], 'pitch': a['pitch'],
               'expanded_mask': b['expanded_mask'][..., :L], 'encoder_attention': a['encoder_attention'],
               'decoder_attention': OrderedDict((k, v[:, :, :L, :L]) for k, v in b['decoder_attention'].items()),
               'expanded_lengths': b['expanded_lengths']}
        return out

    def _duration_mask(self, encoded_text, per_symbol, fill):
        """[B, T] fp32: `fill` everywhere, the given bound at the positions of the listed symbols.  Without per-symbol
        bounds (the usual call) the mask is a constant: one cached device tensor per shape, no host round trip of
        the token ids (three copies and three host waits per call otherwise - 0.25 ms of a 1.2 ms batch-1 predict)."""
        if not per_symbol:
            key = (tuple(encoded_text.shape), fill)
            m = self._const_masks.pop(key, None)
            if m is None:
                m = torch.full(tuple(encoded_text.shape), fill, dtype=torch.float32, device=self.device)
                while len(self._const_masks) >= 32:              # bounded: least recently used shape goes
                    self._const_masks.pop(next(iter(self._const_masks)))
            self._const_masks[key] = m                           # most recently used last
            return m
        np_text = encoded_text.cpu().numpy()
        new_mask = np.full(np_text.shape, fill, dtype=np.float64)
        for sym, val in per_symbol.items():
            new_mask[np_text == self.text_pipeline.tokenizer(sym)[0]] = val
        return torch.from_numpy(new_mask.astype(np.float32)).to(self.device)

    def _make_max_duration_mask(self, encoded_text, phoneme_max_duration):           # :579-586
        return self._duration_mask(encoded_text, phoneme_max_duration, float('inf'))

    def _make_min_duration_mask(self, encoded_text, phoneme_min_duration):           # :588-595
        return self._duration_mask(encoded_text, phoneme_min_duration, 0.0)

    def build_model_weights(self) -> None:                                           # :597-598
        pass    # variables exist from construction; kept for call-site compatibility

    # ------------------------------------------------------------------ persistence (models.py:600-642)
    def save_model(self, path: str):
        """config.yaml + `model_weights.hdf5` in the Keras H5 layout the reference's `load_model` reads
        (`model/models.py:600-618`; written by `model/keras_weights.py` - no h5py needed), plus the Adam
        state, which the reference keeps in its tf.train.Checkpoint instead."""
        from .keras_weights import save_keras_weights
        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        cfg = {k: v for k, v in self.config.items() if k != 'device'}
        cfg.update({'alphabet': ''.join(self.symbols), 'step': self.step})
        try:
            cfg['git_hash'] = subprocess.check_output(['git', 'describe', '--always'],
                                                      stderr=subprocess.DEVNULL).strip().decode()
        except Exception:
            pass
        with open(path / 'config.yaml', 'w') as f:
            yaml.safe_dump(cfg, f)
        save_keras_weights(path / 'model_weights.hdf5', self.weights_dict(), self.config,
                           self.text_pipeline.tokenizer.vocab_size)
        torch.save({'m': self.params.m.cpu(), 'v': self.params.v.cpu(), 'step': self.step,
                    'lr': float(self.lr_dev.item())}, path / 'optimizer.pt')

    @classmethod
    def load_model(cls, path, **kwargs):
        path = Path(path)
        with open(path / 'config.yaml', 'r') as f:
            config = yaml.safe_load(f)
        for k in ('alphabet', 'step', 'git_hash'):
            config.pop(k, None)
        config.update(kwargs)
        model = cls.from_config(config)
        weights = path / 'model_weights.hdf5'
        model.load_weights(weights if weights.exists() else path / 'model_weights.npz')
        opt = path / 'optimizer.pt'
        if opt.exists():
            st = torch.load(opt, map_location='cpu', weights_only=True)
            for key in ('m', 'v'):
                if st[key].numel() != model.params.total:
                    raise ValueError(f'{opt}: Adam state `{key}` has {st[key].numel()} elements, this model\'s flat '
                                     f'parameter buffer has {model.params.total} (different config or layout version)')
            model.params.m.copy_(st['m'])
            model.params.v.copy_(st['v'])
            model._host_step = int(st['step'])
            model.step_dev.fill_(int(st['step']))
            model.lr_dev.fill_(float(st['lr']))
        return model

    def load_weights(self, path):
        """`.hdf5` / `.h5`: a Keras `save_weights` file of the reference model with this config (loaded by
        position with shape checks, exactly as Keras does); `.npz`: this package's variable names."""
        path = str(path)
        if path.endswith('.hdf5') or path.endswith('.h5'):
            from .keras_weights import load_keras_weights
            self.load_weights_dict(load_keras_weights(path, self.config, self.text_pipeline.tokenizer.vocab_size))
            return
        with np.load(path) as z:
            self.load_weights_dict({k: z[k] for k in z.files})

    @classmethod
    def from_config(cls, config: dict, custom_objects=None):                         # :640-642
        return cls(**config)
