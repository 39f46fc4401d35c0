"""Host-side plumbing over the C ABI: torch tensors in, device pointers + the current HIP stream
out.  PyTorch is used for device memory, streams and autograd bookkeeping only - every FLOP below
runs in libttsmi.so (hand-written HIP for gfx950).  Nothing here has a CPU or eager-torch fallback.

Each ``torch.autograd.Function`` is one fused layer of the reference graph (model/layers.py); its
backward calls the matching dgrad/wgrad/bwd entry points.  Parameter gradients can be written
STRAIGHT into a caller-provided gradient buffer (``g*`` arguments, views of the flat gradient
buffer that the single RCCL all-reduce and the fused Adam consume); the Function then returns None
for that parameter."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib
from ._lib import TTSMI_F32, check

LN_EPS = 1e-6


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


# The stream every launch goes to.  Looking it up (torch.cuda.current_stream()) costs ~1 us and a step makes
# ~600 launches from a host loop that is as long as the GPU step itself, so the model pins the handle for the
# duration of a step (pinned_stream); the weight-gradient side stream swaps it while it is current.  Unpinned
# (direct use of the ops), every launch asks torch.
_PINNED_STREAM = None


def _stream():
    h = _PINNED_STREAM
    return h if h is not None else torch.cuda.current_stream().cuda_stream


# torch.cuda.current_stream() builds a Stream object behind three device-index helpers (~9 us a call, ~28 calls per train
# step through Event.record() / `with torch.cuda.stream(...)`).  While a stream is pinned it IS the current stream (every
# place that pins one also makes it current, and autograd replays a backward node on the stream its forward ran on, which is
# the handle the node pins), so the object can be remembered by (handle, device).
_STREAM_OBJS = {}


def _remember_stream(s):
    if len(_STREAM_OBJS) > 64:
        _STREAM_OBJS.clear()
    _STREAM_OBJS[(s.cuda_stream, s.device.index)] = s
    return s


def cur_stream():
    """The current torch.cuda.Stream - without asking torch when a stream is pinned."""
    h = _PINNED_STREAM
    if h is not None:
        s = _STREAM_OBJS.get((h, torch.cuda.current_device()))
        if s is not None:
            return s
    return torch.cuda.current_stream()


class on_stream:
    """`with ops.on_stream(side):` = `with torch.cuda.stream(side), ops.pin_stream(side.cuda_stream):` for a stream of the
    current device, without the two current_stream() look-ups of torch's context manager."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        global _PINNED_STREAM
        self.prev_obj = cur_stream()
        self.prev = _PINNED_STREAM
        _remember_stream(self.stream)
        torch.cuda.set_stream(self.stream)
        _PINNED_STREAM = self.stream.cuda_stream
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self.prev
        torch.cuda.set_stream(self.prev_obj)
        return False


# LayerNorm parameter gradients of a training step, finished in one launch.  Inside `ln_param_batch()` the
# LayerNorm backward calls that write into caller-owned gradient sinks leave their per-workgroup partial sums in
# their workspaces (kept alive here) and `ln_flush()` reduces all of them with ONE kernel
# (ttsmi_layernorm_param_reduce_batched) - ~30 launches of ~6.7 us less on the critical path of a step.
# Everything is on the main stream, in program order: nothing to synchronise, results unchanged bit for bit.
# Outside the context (direct use of the ops, unit tests) every call reduces immediately.
_LN_PENDING = None
_LN_BATCHED = os.environ.get('TTSMI_LN_BATCHED', '1') != '0'       # measurement knob: 0 = one reduce launch per call


class ln_param_batch:
    def __init__(self, enabled: bool = True):
        self.enabled = bool(enabled) and _LN_BATCHED

    def __enter__(self):
        global _LN_PENDING
        self.prev = _LN_PENDING
        if self.enabled:
            _LN_PENDING = []
        return self

    def __exit__(self, exc_type, *exc):
        global _LN_PENDING
        try:
            if exc_type is None:
                ln_flush()
        finally:
            _LN_PENDING = self.prev
        return False


def _ln_defer(ws, dgamma, dbeta, dps, M, C, nparts=None, stream=None) -> None:
    """nparts: partial rows in `ws` (default: what ttsmi_add_layernorm_bwd leaves for M rows); stream: the raw stream
    the partial sums are produced on (default: the current launch stream)."""
    if nparts is None:
        nparts = _lib.lib().ttsmi_add_layernorm_bwd_nparts(int(M))
    _LN_PENDING.append((ws, dgamma, dbeta, dps, int(nparts), int(C), _stream() if stream is None else stream))


def ln_flush(only_this_stream: bool = False) -> None:
    """Reduce every pending LayerNorm parameter gradient (no-op when nothing is pending): ONE launch per stream that
    produced partial sums, issued on THAT stream (program order behind its partials - the caller joins the streams as
    it does for the weight gradients).  only_this_stream: only the entries of the current launch stream."""
    if not _LN_PENDING:
        return
    cur = _stream()
    groups, rest = {}, []
    for e in _LN_PENDING:
        if only_this_stream and e[6] != cur:
            rest.append(e)
        else:
            groups.setdefault(e[6], []).append(e)
    for handle, pend in groups.items():
        n = len(pend)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
        ws = PA(*[e[0].data_ptr() for e in pend])
        dg = PA(*[e[1].data_ptr() for e in pend])
        db = PA(*[e[2].data_ptr() for e in pend])
        ds = PA(*[(e[3].data_ptr() if e[3] is not None else None) for e in pend])
        nw, Cs = IA(*[e[4] for e in pend]), IA(*[e[5] for e in pend])
        check(_lib.lib().ttsmi_layernorm_param_reduce_batched_nw(ctypes.addressof(ws), ctypes.addressof(dg),
                                                                 ctypes.addressof(db), ctypes.addressof(ds),
                                                                 ctypes.addressof(nw), ctypes.addressof(Cs), n, handle),
              'layernorm_param_reduce_batched')
    _LN_PENDING[:] = rest             # the reduced workspaces go back to the allocator, in stream order


class pinned_stream:
    """`with ops.pinned_stream():` - launches inside go to the stream that is current on entry."""

    def __enter__(self):
        global _PINNED_STREAM
        self.prev = _PINNED_STREAM
        _PINNED_STREAM = _remember_stream(torch.cuda.current_stream()).cuda_stream
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self.prev
        return False


class pin_stream:
    """`with ops.pin_stream(handle):` - launches inside go to the given raw hipStream_t (None: ask torch)."""

    def __init__(self, handle):
        self.handle = handle

    def __enter__(self):
        global _PINNED_STREAM
        self.prev = _PINNED_STREAM
        _PINNED_STREAM = self.handle
        return self

    def __exit__(self, *exc):
        global _PINNED_STREAM
        _PINNED_STREAM = self.prev
        return False


def _ws(nbytes: int, device) -> torch.Tensor:
    """A byte workspace.  Sized in whole 256-byte units: several entry points require 256-byte aligned workspaces, which an
    allocator that places tensors flush against the END of a mapping (tests/guard_alloc.cpp) only gives to such sizes."""
    return torch.empty((int(nbytes) + 255) // 256 * 256, dtype=torch.uint8, device=device)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise _lib.TtsmiError('libttsmi ops need CUDA(HIP) tensors; there is no CPU fallback')


class DropCtx:
    """Dropout stream of one model: host seed + device step counter (graph-replay safe) and a
    running site id so every dropout site of the graph draws from its own stream."""

    def __init__(self, seed: int = 0, step_dev: Optional[torch.Tensor] = None):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.step_dev = step_dev
        self._site = 0

    def reset(self):
        self._site = 0

    def site(self) -> int:
        self._site += 1
        return self._site


# =================================================================================================
# raw calls
# =================================================================================================
# dtype of the fp32-tensor GEMM family (ttsmi_linear_* / ttsmi_conv1d_*) when a caller does not name one: TTSMI_F32 (exact-fp32
# MFMA) or TTSMI_BF16X3 (three bf16 MFMAs per product).  A model sets it at the top of every public call (precision='bf16x3');
# the autograd nodes that run inside that call read it at launch time.
F32_GEMM_DTYPE = TTSMI_F32


def set_f32_gemm_dtype(dtype):
    global F32_GEMM_DTYPE
    F32_GEMM_DTYPE = int(dtype)


def linear_fwd(x, w, b, relu=False, x2=None, out=None, dtype=None):
    dtype = F32_GEMM_DTYPE if dtype is None else dtype
    _need_gpu(x)
    M, K1 = x.shape
    K, N = w.shape
    if x2 is not None:
        assert x2.shape[0] == M and K1 + x2.shape[1] == K
    else:
        assert K1 == K, (x.shape, w.shape)
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(_lib.lib().ttsmi_linear_fwd(_p(x), x.stride(0), _p(x2), 0 if x2 is None else x2.stride(0),
                                      K1 if x2 is not None else 0, _p(w), w.stride(0), _p(b), _p(y),
                                      y.stride(0), M, N, K, int(relu), dtype, _stream()), 'linear_fwd')
    return y


def linear_dgrad(dy, w, relu_src=None, out=None, dtype=None, accumulate=False):
    """dx = dy . w^T (* relu mask); accumulate=True adds into `out` instead of overwriting it."""
    dtype = F32_GEMM_DTYPE if dtype is None else dtype
    M, N = dy.shape
    K = w.shape[0]
    assert not accumulate or out is not None
    dx = out if out is not None else torch.empty((M, K), dtype=torch.float32, device=dy.device)
    check(_lib.lib().ttsmi_linear_dgrad(_p(dy), dy.stride(0), _p(w), w.stride(0), _p(relu_src),
                                        0 if relu_src is None else relu_src.stride(0), _p(dx),
                                        dx.stride(0), M, N, K, int(accumulate), dtype, _stream()), 'linear_dgrad')
    return dx


def linear_wgrad(x, dy, dw, db, dtype=None):
    dtype = F32_GEMM_DTYPE if dtype is None else dtype
    M, K = x.shape
    N = dy.shape[1]
    l = _lib.lib()
    ws = _ws(l.ttsmi_linear_wgrad_ws_bytes(M, N, K), x.device)
    check(l.ttsmi_linear_wgrad(_p(x), x.stride(0), _p(dy), dy.stride(0), _p(dw), dw.stride(0), _p(db),
                               M, N, K, _p(ws), ws.numel(), dtype, _stream()), 'linear_wgrad')


def conv1d_fwd(x, w, b, relu=False, dtype=None):
    dtype = F32_GEMM_DTYPE if dtype is None else dtype
    B, T, Cin = x.shape
    k, _, Cout = w.shape
    y = torch.empty((B, T, Cout), dtype=torch.float32, device=x.device)
    check(_lib.lib().ttsmi_conv1d_fwd(_p(x), _p(w), _p(b), _p(y), B, T, Cin, Cout, k, int(relu), dtype,
                                      _stream()), 'conv1d_fwd')
    return y


def conv1d_dgrad(dy, w, relu_src=None, dtype=None):
    dtype = F32_GEMM_DTYPE if dtype is None else dtype
    B, T, Cout = dy.shape
    k, Cin, _ = w.shape
    dx = torch.empty((B, T, Cin), dtype=torch.float32, device=dy.device)
    check(_lib.lib().ttsmi_conv1d_dgrad(_p(dy), _p(w), _p(relu_src), _p(dx), B, T, Cin, Cout, k, dtype,
                                        _stream()), 'conv1d_dgrad')
    return dx


def conv1d_wgrad(x, dy, dw, db, dtype=None):
    dtype = F32_GEMM_DTYPE if dtype is None else dtype
    B, T, Cin = x.shape
    Cout = dy.shape[2]
    k = dw.shape[0]
    l = _lib.lib()
    ws = _ws(l.ttsmi_conv1d_wgrad_ws_bytes(B, T, Cin, Cout, k), x.device)
    check(l.ttsmi_conv1d_wgrad(_p(x), _p(dy), _p(dw), _p(db), B, T, Cin, Cout, k, _p(ws), ws.numel(),
                               dtype, _stream()), 'conv1d_wgrad')


def token_pad_mask(tokens):
    B, T = tokens.shape
    pad = torch.empty((B, T), dtype=torch.uint8, device=tokens.device)
    klen = torch.empty((B,), dtype=torch.int32, device=tokens.device)
    check(_lib.lib().ttsmi_token_pad_mask(_p(tokens), _p(pad), _p(klen), B, T, _stream()), 'token_pad_mask')
    return pad, klen


def length_pad_mask(lens, T):
    B = lens.shape[0]
    pad = torch.empty((B, T), dtype=torch.uint8, device=lens.device)
    klen = torch.empty((B,), dtype=torch.int32, device=lens.device)
    check(_lib.lib().ttsmi_length_pad_mask(_p(lens), _p(pad), _p(klen), B, T, _stream()), 'length_pad_mask')
    return pad, klen


def lenreg_index(dur, cap):
    """dur [B,Tp] int32 or float32 -> (idx [B,cap] i32, cum [B,Tp+1] i32, len [B] i32)."""
    _need_gpu(dur)
    dur = _c(dur)
    B, Tp = dur.shape
    assert dur.dtype in (torch.int32, torch.float32), dur.dtype
    idx = torch.empty((B, cap), dtype=torch.int32, device=dur.device)
    cum = torch.empty((B, Tp + 1), dtype=torch.int32, device=dur.device)
    ln = torch.empty((B,), dtype=torch.int32, device=dur.device)
    check(_lib.lib().ttsmi_lenreg_index(_p(dur), int(dur.dtype == torch.int32), _p(idx), _p(cum), _p(ln),
                                        B, Tp, cap, _stream()), 'lenreg_index')
    return idx, cum, ln


def attention_weights(qkv, key_pad, lse, B, H, T, dh, p_drop=0.0, drop: Optional[DropCtx] = None,
                      site=0, dtype=TTSMI_F32, dmask=None, out=None):
    """dmask: the layer's keep-bit table (bf16 tensors only) - the same decisions as the hash, one bit test per weight.
    out: a caller-owned [B,H,T,T] fp32 buffer (the train step's ring of map buffers, models.py:_map_buffer)."""
    if out is not None:
        assert out.shape == (B, H, T, T) and out.dtype == torch.float32 and out.is_contiguous()
    w = out if out is not None else torch.empty((B, H, T, T), dtype=torch.float32, device=qkv.device)
    if dmask is not None and p_drop > 0 and dtype == _lib.TTSMI_BF16_IO:
        check(_lib.lib().ttsmi_attention_weights_masked(_p(qkv), _p(key_pad), _p(lse), _p(w), B, H, T, dh, float(p_drop),
                                                        _p(dmask), dtype, _stream()), 'attention_weights_masked')
        return w
    check(_lib.lib().ttsmi_attention_weights(_p(qkv), _p(key_pad), _p(lse), _p(w), B, H, T, dh,
                                             float(p_drop), drop.seed if drop else 0,
                                             _p(drop.step_dev) if drop else None, site, dtype,
                                             _stream()), 'attention_weights')
    return w


_CONV_PLAIN = os.environ.get('TTSMI_CONV_PLAIN', '1') != '0'        # bf16 conv stacks as plain GEMMs over a zero-margin layout (ConvStackFn)
_ATTN_DROPBITS = os.environ.get('TTSMI_ATTN_DROPBITS', '1') != '0'    # measurement knob: 0 = hash in the inner loops


def attention_dropmask(B, H, T, p_drop, drop: DropCtx, site, device, out=None):
    """Keep-bit table of one attention layer for this step (include/ttsmi.h: ttsmi_attention_dropmask)."""
    l = _lib.lib()
    mask = out if out is not None else torch.empty(int(l.ttsmi_attention_dropmask_bytes(B, H, T)), dtype=torch.uint8,
                                                    device=device)
    check(l.ttsmi_attention_dropmask(_p(mask), B, H, T, float(p_drop), drop.seed, _p(drop.step_dev), int(site),
                                     _stream()), 'attention_dropmask')
    return mask


def adam_tf(p, g, m, v, lr_dev, step_dev, b1=0.9, b2=0.98, eps=1e-9, shadow=None):
    check(_lib.lib().ttsmi_adam_tf(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_dev), _p(step_dev),
                                   b1, b2, eps, _p(shadow), _stream()), 'adam_tf')


def step_increment(step_dev):
    check(_lib.lib().ttsmi_step_increment(_p(step_dev), _stream()), 'step_increment')


def stft_logmel(wav, clip_off, frame_off, total_frames, n_fft, hop, window, n_mels, mel_lo, mel_cnt,
                mel_ptr, mel_w, normalizer, clip_min, out=None):
    if out is None:
        out = torch.empty((int(total_frames), n_mels), dtype=torch.float32, device=wav.device)
    check(_lib.lib().ttsmi_stft_logmel(_p(wav), _p(clip_off), _p(frame_off), clip_off.numel() - 1,
                                       int(total_frames), n_fft, hop, _p(window), n_mels, _p(mel_lo),
                                       _p(mel_cnt), _p(mel_ptr), _p(mel_w), normalizer, clip_min, _p(out),
                                       _stream()), 'stft_logmel')
    return out


def mel_nnls(mel, pinv_t, mel_lo, mel_cnt, mel_ptr, mel_w, inv_lipschitz, n_iter=512, power=1.0):
    """ttsmi_mel_nnls (include/ttsmi.h): amplitude mel [T, n_mels] fp32 -> non-negative linear magnitudes [T, n_bins] fp32
    (frame-major, what `griffinlim` takes) whose mel projection is closest to it."""
    T, n_mels = int(mel.shape[0]), int(mel.shape[1])
    n_bins = int(pinv_t.shape[1])
    assert mel.is_contiguous() and mel.dtype == torch.float32 and tuple(pinv_t.shape) == (n_mels, n_bins)
    x = torch.empty((T, n_bins), dtype=torch.float32, device=mel.device)
    check(_lib.lib().ttsmi_mel_nnls(_p(mel), _p(pinv_t), _p(mel_lo), _p(mel_cnt), _p(mel_ptr), _p(mel_w), mel_w.numel(),
                                    _p(x), T, n_mels, n_bins, float(inv_lipschitz), int(n_iter), 1.0 / float(power),
                                    _stream()), 'mel_nnls')
    return x


def griffinlim(mag, angles, window, wss, n_fft, hop, n_iter, momentum):
    """ttsmi_griffinlim (include/ttsmi.h): mag [T,513] fp32, angles [T,513,2] fp32 (updated in place), window [n_fft],
    wss [n_fft + hop (T - 1)] -> wav [hop (T - 1)] fp32 on the device."""
    l = _lib.lib()
    T = int(mag.shape[0])
    assert mag.is_contiguous() and angles.is_contiguous() and tuple(angles.shape) == (T, mag.shape[1], 2)
    assert wss.numel() == n_fft + hop * (T - 1)
    wav = torch.empty(hop * (T - 1), dtype=torch.float32, device=mag.device)
    ws = _ws(max(int(l.ttsmi_griffinlim_ws_bytes(T)), 256), mag.device)
    check(l.ttsmi_griffinlim(_p(mag), _p(angles), _p(window), _p(wss), T, n_fft, hop, int(n_iter), float(momentum),
                             _p(wav), _p(ws), ws.numel(), _stream()), 'griffinlim')
    return wav


class Shadow:
    """bf16 copies of one GEMM weight for the TTSMI_BF16 path, refreshed after every optimiser step:
      wb  - the weight as stored, bf16 ([K,N] Dense; dgrad reads its rows as K-contiguous operands)
      wt  - its transpose, bf16 ([N,K] Dense / [Cout, k*Cin] Conv1D; the forward B operand)
      wd  - Conv1D only: dgrad operand [Cin, k*pad8(Cout)] with flipped taps (zero pad columns)."""
    __slots__ = ('wb', 'wt', 'wd')

    def __init__(self, wb=None, wt=None, wd=None):
        self.wb, self.wt, self.wd = wb, wt, wd


def _al(t, n):
    return t is not None and t.data_ptr() % 16 == 0 and t.stride(0) % n == 0


def hgemm_tn(a, b_h, bias=None, relu=False, a2=None, relu_src=None, conv=None, rows=None, out=None,
             accumulate=False, out_bf16=False):
    """c[M,N] (+)= act(sum_k a[m,k]*b_h[n,k] + bias) * (relu_src > 0) on bf16 MFMA (include/ttsmi.h).
    a may be fp32 or bf16; relu_src may be fp32 or bf16; out_bf16 stores c as bf16."""
    a_f32 = a.dtype == torch.float32
    N, K = b_h.shape
    M = a.shape[0] if rows is None else rows
    K1 = a.shape[1] if a2 is not None else 0
    assert not accumulate or out is not None
    c = out if out is not None else torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32,
                                                device=a.device)
    taps, T, C, pad = conv if conv is not None else (1, 0, 0, 0)
    flags = (1 if relu else 0) | (2 if accumulate else 0) | (4 if c.dtype == torch.bfloat16 else 0) | \
            (8 if (relu_src is not None and relu_src.dtype == torch.bfloat16) else 0)
    check(_lib.lib().ttsmi_hgemm_tn(_p(a), int(a_f32), a.stride(0), _p(a2), 0 if a2 is None else a2.stride(0),
                                    K1, _p(b_h), b_h.stride(0), _p(bias), _p(relu_src),
                                    0 if relu_src is None else relu_src.stride(0), _p(c), c.stride(0), M, N, K,
                                    flags, taps, T, C, pad, _stream()), 'hgemm_tn')
    return c


def cast_transpose_bf16(src2d, taps=1, T=0, pad=0):
    """fp32 [R,C] -> bf16 [taps*C, ldt] (ldt = R rounded up to 8, zero tail)."""
    R, C = src2d.shape
    ldt = (R + 7) // 8 * 8
    dst = torch.empty((taps * C, ldt), dtype=torch.bfloat16, device=src2d.device)
    check(_lib.lib().ttsmi_cast_transpose_bf16(_p(src2d), src2d.stride(0), _p(dst), ldt, R, C, taps, T, pad,
                                               _stream()), 'cast_transpose_bf16')
    return dst


def hgemm_wgrad(xT, dyT, dw, db, rows):
    kin, ldt = xT.shape
    n = dyT.shape[0]
    l = _lib.lib()
    ws = _ws(l.ttsmi_hgemm_wgrad_ws_bytes(rows, kin, n), xT.device)
    check(l.ttsmi_hgemm_wgrad(_p(xT), _p(dyT), ldt, _p(dw), dw.stride(0), _p(db), rows, kin, n, _p(ws),
                              ws.numel(), _stream()), 'hgemm_wgrad')


def _pad8(n):
    return (n + 7) // 8 * 8


def conv_wdgrad_layout_bf16(w):
    """[Cin, k*pad8(Cout)] bf16 dgrad operand of a Conv1D weight [k, Cin, Cout] (zero pad columns)."""
    k, cin, cout = w.shape
    dst = torch.empty((cin, k * _pad8(cout)), dtype=torch.bfloat16, device=w.device)
    check(_lib.lib().ttsmi_conv_wdgrad_layout_bf16(_p(w), _p(dst), k, cin, cout, _pad8(cout), _stream()),
          'conv_wdgrad_layout')
    return dst


def make_shadow(w):
    """Build the bf16 shadow of a Dense [K,N] or Conv1D [k,Cin,Cout] weight (fp32 tensor)."""
    with torch.no_grad():
        wf = w.detach()
        if wf.dim() == 2:
            wb = torch.empty(wf.shape, dtype=torch.bfloat16, device=wf.device)
            check(_lib.lib().ttsmi_cast_f32_to_bf16(_p(wf), _p(wb), wf.numel(), _stream()), 'cast_f32_to_bf16')
            return Shadow(wb=wb, wt=cast_transpose_bf16(wf))
        k, cin, cout = wf.shape
        return Shadow(wt=cast_transpose_bf16(wf.reshape(k * cin, cout)), wd=conv_wdgrad_layout_bf16(wf))


def refresh_shadow(sh, w):
    """Re-cast in place (same buffers: safe under hipGraph replay)."""
    wf = w.detach()
    l = _lib.lib()
    if wf.dim() == 2:
        check(l.ttsmi_cast_f32_to_bf16(_p(wf), _p(sh.wb), wf.numel(), _stream()), 'cast_f32_to_bf16')
        K, N = wf.shape
        check(l.ttsmi_cast_transpose_bf16(_p(wf), wf.stride(0), _p(sh.wt), sh.wt.stride(0), K, N, 1, 0, 0,
                                          _stream()), 'cast_transpose_bf16')
    else:
        k, cin, cout = wf.shape
        check(l.ttsmi_cast_transpose_bf16(_p(wf), cout, _p(sh.wt), sh.wt.stride(0), k * cin, cout, 1, 0, 0,
                                          _stream()), 'cast_transpose_bf16')
        check(l.ttsmi_conv_wdgrad_layout_bf16(_p(wf), _p(sh.wd), k, cin, cout, _pad8(cout), _stream()),
              'conv_wdgrad_layout')


class ShadowSet:
    """All bf16 weight shadows of a model, refreshed with O(1) launches per optimiser step:
      * `wb` (weights as stored) are views of ONE flat bf16 buffer that the fused Adam kernel writes
        while it updates the fp32 master weights (no extra launch, no extra read of the weights);
      * every `wt` (W^T / conv forward layout) comes from ONE batched cast-transpose launch driven by
        a descriptor table that lives on the device;
      * the few Conv1D dgrad layouts (`wd`) keep one small launch each.
    flat: the fp32 flat parameter buffer; offsets: name -> (offset, numel); views: name -> fp32 view."""

    def __init__(self, flat, offsets, views, names):
        import numpy as np
        dev = flat.device
        self.flat = flat
        self.flat_bf16 = torch.empty(flat.numel(), dtype=torch.bfloat16, device=dev)
        self.sh = {}
        self._conv = []
        recs = []
        tile = 0
        for name in names:
            w = views[name].detach()
            o, n = offsets[name]
            assert (o * 2) % 16 == 0, 'parameter offsets must keep bf16 views 16-byte aligned'
            if w.dim() == 2:
                K, N = w.shape
                wt = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
                self.sh[name] = Shadow(wb=self.flat_bf16[o:o + n].view(K, N), wt=wt)
                R, C = K, N
            else:
                k, cin, cout = w.shape
                wt = torch.empty((cout, k * cin), dtype=torch.bfloat16, device=dev)
                wd = torch.empty((cin, k * _pad8(cout)), dtype=torch.bfloat16, device=dev)
                self.sh[name] = Shadow(wt=wt, wd=wd)
                self._conv.append((w, wd))
                R, C = k * cin, cout
            tiles_r = (R + 63) // 64
            recs.append((w.data_ptr(), wt.data_ptr(), C, R, R, C, tile, tiles_r))
            tile += tiles_r * ((C + 63) // 64)
        self.n_desc, self.total_tiles = len(recs), tile
        dt = np.dtype([('src', '<u8'), ('dst', '<u8'), ('ld_src', '<i8'), ('ld_dst', '<i8'), ('R', '<i4'),
                       ('C', '<i4'), ('tile_start', '<i4'), ('tiles_r', '<i4')])
        arr = np.array(recs, dtype=dt)
        assert arr.itemsize == 48
        self.desc = torch.from_numpy(arr.view(np.uint8).copy()).to(dev) if recs else None

    def refresh(self, wb_is_current: bool):
        """wb_is_current: the Adam kernel of this step already wrote the flat bf16 copy."""
        l = _lib.lib()
        if not wb_is_current:
            check(l.ttsmi_cast_f32_to_bf16(_p(self.flat), _p(self.flat_bf16), self.flat.numel(), _stream()),
                  'cast_f32_to_bf16')
        if self.desc is not None:
            check(l.ttsmi_cast_transpose_bf16_batched(_p(self.desc), self.n_desc, self.total_tiles, _stream()),
                  'cast_transpose_bf16_batched')
        for w, wd in self._conv:
            k, cin, cout = w.shape
            check(l.ttsmi_conv_wdgrad_layout_bf16(_p(w), _p(wd), k, cin, cout, _pad8(cout), _stream()),
                  'conv_wdgrad_layout')


# ---- precision-dispatching wrappers: sh is None -> exact-fp32 MFMA kernels ------------------------
def dense_fwd(x, w, b, relu, x2, sh):
    K = w.shape[0]
    if sh is not None and K % 8 == 0 and _al(x, 4) and (x2 is None or (_al(x2, 4) and x.shape[1] % 8 == 0)):
        return hgemm_tn(x, sh.wt, b, relu, x2)
    return linear_fwd(x, w, b, relu, x2)


def dense_dgrad(dy, w, sh, k0, k1, relu_src=None, out=None, accumulate=False):
    """dx (+)= dy . w[k0:k1]^T (* relu mask)."""
    N = w.shape[1]
    if sh is not None and N % 8 == 0 and _al(dy, 4) and k0 % 8 == 0:
        return hgemm_tn(dy, sh.wb[k0:k1], None, False, None, relu_src, out=out, accumulate=accumulate)
    return linear_dgrad(dy, w[k0:k1], relu_src, out=out, accumulate=accumulate)


class _WgradState:
    __slots__ = ('stream', 'handle', 'ws', 'pending', 'keep')

    def __init__(self):
        self.stream = None          # the side stream (created on first use, on the device that is current then)
        self.handle = None          # raw hipStream_t of `stream`
        self.ws = None              # persistent split-K workspace of the side stream (its kernels run in order)
        self.pending = False
        self.keep = []


class _WgradStream:
    """Weight gradients are off the critical path of backward (nothing downstream reads them until
    the optimiser step), so they are launched on a second HIP stream and overlap the dgrad / attention
    / LayerNorm chain of the main stream: both are HBM-latency bound and fill each other's bubbles.
    Joined (wgrad_join) before the gradient all-reduce / Adam.  One state per device: a stream belongs to
    the device it was created on, so models on different GPUs of one process never share one."""
    enabled = False
    per_device = {}

    @classmethod
    def cur(cls) -> _WgradState:
        """The weight-gradient stream's state on the current device.  (Round 5 measured a second stream for the encoder
        stack's blocks - level to slightly slower, profiles/r05_wgrad_lanes_ab.txt - and round 6 removed it.)"""
        key = torch.cuda.current_device()
        st = cls.per_device.get(key)
        if st is None:
            st = cls.per_device[key] = _WgradState()
        return st

    @classmethod
    def lanes(cls):
        st = cls.per_device.get(torch.cuda.current_device())
        return [st] if st is not None else []


def _new_wgrad_stream():
    """The weight-gradient side stream.  (Round 6 measured it confined to a CU mask - hipExtStreamCreateWithCUMask through
    ttsmi_debug_stream_create_cu_mask, tools/probe_cu_mask.py: no mask selects whole XCDs on this part and any mask costs the
    step 72 %, profiles/r06_wgrad_cu_mask_ab.txt - so it is an ordinary stream.)"""
    return torch.cuda.Stream(priority=_WGRAD_PRIO)


def enable_wgrad_stream(flag: bool = True):
    _WgradStream.enabled = bool(flag)


def _on_wgrad_stream(fn, *inputs):
    if not _WgradStream.enabled:
        return fn()
    W = _WgradStream.cur()
    main = cur_stream()
    if W.stream is None:
        W.stream = _new_wgrad_stream()
    side = W.stream
    side.wait_stream(main)                    # the operands were produced on the main stream
    if _PINNED_STREAM is not None:
        with on_stream(side):
            fn()                              # workspace allocated in here belongs to the side stream
    else:
        with torch.cuda.stream(side):
            fn()
    # Hold a reference until the join (which makes the main stream wait for the side stream):
    #  * the caching allocator cannot recycle the operands while the side stream may still read them
    #    (no record_stream needed, which also keeps this legal under hipGraph capture);
    #  * autograd sums the gradients of a multiply-used tensor IN PLACE into a buffer it owns
    #    exclusively (use_count == 1) - e.g. the LayerNorm backward output that is both this layer's dy
    #    and the residual branch's gradient.  A second owner makes it allocate the sum instead of
    #    mutating a tensor the side stream is still reading.
    W.keep.extend(inputs)
    W.pending = True


_WGRAD_GENERIC = os.environ.get('TTSMI_WGRAD_GENERIC', '0') == '1'      # measurement knob: old submission path


def wgrad_rows_async(x, dy, dw, db, conv=None):
    """hgemm_wgrad_rows on the weight-gradient stream without entering a torch stream context: the
    launch takes the stream handle explicitly and the split-K workspace is one persistent buffer owned by
    that stream (its kernels execute in order, so reuse is safe) - a third of the host cost of the generic
    _on_wgrad_stream path, which matters with ~60 weight gradients per step."""
    if not _WgradStream.enabled:
        return hgemm_wgrad_rows(x, dy, dw, db, conv)
    W = _WgradStream.cur()
    if _WGRAD_GENERIC:
        return _on_wgrad_stream(lambda: hgemm_wgrad_rows(x, dy, dw, db, conv), x, dy)
    if W.stream is None:
        W.stream = _new_wgrad_stream()
    if W.handle is None:
        W.handle = W.stream.cuda_stream
    ev = torch.cuda.Event()
    ev.record(cur_stream())                   # on the current (main) stream: the operands are complete here
    W.stream.wait_event(ev)
    M, N = x.shape[0], dy.shape[1]
    taps, T, C, pad = conv if conv is not None else (1, 0, 0, 0)
    kin = dw.shape[0]
    l = _lib.lib()
    need = l.ttsmi_hgemm_wgrad_rows_ws_bytes(M, kin, N)
    if W.ws is None or W.ws.numel() < need:
        with torch.cuda.stream(W.stream):
            W.ws = torch.empty(int(max(need, 1 << 26)), dtype=torch.uint8, device=x.device)
    check(l.ttsmi_hgemm_wgrad_rows(_p(x), int(x.dtype == torch.bfloat16), x.stride(0), _p(dy),
                                   int(dy.dtype == torch.bfloat16), dy.stride(0), _p(dw), dw.stride(0), _p(db), M,
                                   kin, N, taps, T, C, pad, _p(W.ws), W.ws.numel(), W.handle), 'hgemm_wgrad_rows')
    W.keep.append(x)
    W.keep.append(dy)
    W.pending = True


def wgrad_join():
    for W in _WgradStream.lanes():
        if W.pending:
            cur_stream().wait_stream(W.stream)
            W.pending = False
        W.keep.clear()


def dense_wgrad(x, dy, dw, db, sh, dyT=None):
    """dw = x^T . dy, db = colsum(dy).  Returns dyT (bf16 path) so a dual-A caller can reuse it."""
    if sh is not None:
        M, K = x.shape
        N = dy.shape[1]
        if K % 4 == 0 and N % 4 == 0 and _al(x, 4) and _al(dy, 4):
            # reads the fp32 rows once, transposes in LDS
            wgrad_rows_async(x, dy, dw, db)
            return None
        if dyT is None:
            dyT = cast_transpose_bf16(dy)
        hgemm_wgrad(cast_transpose_bf16(x), dyT, dw, db, M)
        return dyT
    _on_wgrad_stream(lambda: linear_wgrad(x, dy, dw, db), x, dy)
    return None


def hgemm_wgrad_rows(x, dy, dw, db, conv=None):
    """dw[K,N] = x[M,K]^T . dy[M,N] (+db) on bf16 MFMA straight from row-major fp32 operands.
    conv = (taps, T, Cin, pad): x is [B*T, Cin], dw is [taps*Cin, N]."""
    M = x.shape[0]
    N = dy.shape[1]
    taps, T, C, pad = conv if conv is not None else (1, 0, 0, 0)
    kin = dw.shape[0]
    l = _lib.lib()
    ws = _ws(l.ttsmi_hgemm_wgrad_rows_ws_bytes(M, kin, N), x.device)
    check(l.ttsmi_hgemm_wgrad_rows(_p(x), int(x.dtype == torch.bfloat16), x.stride(0), _p(dy),
                                   int(dy.dtype == torch.bfloat16), dy.stride(0), _p(dw), dw.stride(0), _p(db), M,
                                   kin, N, taps, T, C, pad, _p(ws), ws.numel(), _stream()), 'hgemm_wgrad_rows')


def _sink(g, like):
    """Gradient destination: the caller's buffer view, or a fresh tensor."""
    return g if g is not None else torch.empty_like(like)


# =================================================================================================
# autograd Functions (one per fused layer)
# =================================================================================================
_GRAD_SINK_CHECK = os.environ.get('TTSMI_GRAD_SINK_CHECK', '1') != '0'


class GradSink:
    """The gradient of a tensor with several consumers on the per-layer path (a block's input feeds the qkv projection, the
    q_in half of the output projection and the residual of res-norm 1; the conv stack's input is also res-norm 2's residual),
    summed WITHOUT autograd's add launches.  The residual's LayerNorm backward runs first among the consumers (it consumes
    what the others produce): it returns its `dres` to autograd as usual and leaves the tensor here; the other consumers
    ADD their contribution into it in place (the dgrad GEMM's accumulate epilogue) and return None.  Autograd keeps the
    first gradient that arrives by reference and calls the producer's backward only after every consumer's has run, in
    stream order - by then the buffer holds the sum.  One object per (tensor, step); never shared across steps.

    That first-gradient-by-reference behaviour of the autograd engine is not a documented contract (a second
    gradient-producing consumer, a cross-stream edge or a future torch would make it sum OUT of place, and the in-place
    adds would land in a dead buffer): `watch(t)` puts a hook on the guarded tensor that sees the gradient autograd
    hands to its producer and raises unless it IS the buffer the consumers added into (TTSMI_GRAD_SINK_CHECK=0 removes
    the hook)."""
    __slots__ = ('buf', 'name')

    def __init__(self, name=''):
        self.buf = None
        self.name = name

    def watch(self, t):
        if _GRAD_SINK_CHECK and t is not None and t.requires_grad:
            t.register_hook(self._check)
        return self

    def _check(self, g):
        if self.buf is not None and (g.data_ptr() != self.buf.data_ptr() or g.numel() != self.buf.numel()):
            raise RuntimeError(f'GradSink {self.name}: autograd summed the gradient out of place (its buffer {g.data_ptr():#x} is '
                               f'not the one the consumers accumulated into, {self.buf.data_ptr():#x}): contributions would '
                               f'be dropped - run with TTSMI_GRAD_SINK=0')
        return None


class LinearFn(torch.autograd.Function):
    """y = [x | x2] . w + b.   Dense at model/layers.py:116-120,148-149; model/models.py:422.
    bf16 plumbing of the per-layer path (conv blocks): out_bf16 stores y as bf16 (the attention kernels' operand);
    x_h is a bf16 copy of x used as the GEMM operand (x itself stays the differentiable input); a bf16 x2 (the attention
    context) receives its gradient as bf16."""

    @staticmethod
    def forward(ctx, x, x2, w, b, gw, gb, sh=None, out_bf16=False, x_h=None, x_sink=None):
        ctx.x_sink = x_sink
        x = _c(x)
        x2 = None if x2 is None else _c(x2)
        xa = x if x_h is None else _c(x_h)
        h_ok = sh is not None and w.shape[0] % 8 == 0 and _al(xa, 4)
        if (out_bf16 or xa.dtype == torch.bfloat16 or (x2 is not None and x2.dtype == torch.bfloat16)):
            assert h_ok and (x2 is None or x2.dtype == xa.dtype), 'bf16 operands need the bf16 MFMA route and one operand type'
            y = hgemm_tn(xa, sh.wt, b, False, x2, out_bf16=out_bf16)
        else:
            y = dense_fwd(xa, w, b, False, x2, sh)
        ctx.save_for_backward(xa, x2, w)
        ctx.sinks = (gw, gb)
        ctx.has_b = b is not None
        ctx.sh = sh
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x2, w = ctx.saved_tensors
        gw, gb = ctx.sinks
        dy = _c(dy)
        K1 = x.shape[1]
        dw = _sink(gw, w)
        db = _sink(gb, w[0]) if ctx.has_b else None
        sh, K = ctx.sh, w.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            sink = ctx.x_sink
            if sink is not None and sink.buf is not None:
                dense_dgrad(dy, w, sh, 0, K1, out=sink.buf.view(dy.shape[0], K1), accumulate=True)     # (GradSink)
            else:
                dx = dense_dgrad(dy, w, sh, 0, K1)
        dyT = dense_wgrad(x, dy, dw[:K1], db, sh)
        dx2 = None
        if x2 is not None:
            if ctx.needs_input_grad[1]:
                if x2.dtype == torch.bfloat16:
                    dx2 = hgemm_tn(dy, sh.wb[K1:K], None, False, None, None, out_bf16=True)
                else:
                    dx2 = dense_dgrad(dy, w, sh, K1, K)
            dense_wgrad(x2, dy, dw[K1:], None, sh, dyT)
        return (dx, dx2, (None if gw is not None else dw), (None if (gb is not None or db is None) else db),
                None, None, None, None, None, None)


class FFNFn(torch.autograd.Function):
    """relu(x.w1+b1).w2+b2 - the two Dense layers of FFNResNorm (model/layers.py:99-100)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, gw1, gb1, gw2, gb2, sh1=None, sh2=None):
        x = _c(x)
        h = dense_fwd(x, w1, b1, True, None, sh1)
        y = dense_fwd(h, w2, b2, False, None, sh2)
        ctx.save_for_backward(x, h, w1, w2)
        ctx.sinks = (gw1, gb1, gw2, gb2)
        ctx.sh = (sh1, sh2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, w1, w2 = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = ctx.sinks
        dy = _c(dy)
        sh1, sh2 = ctx.sh
        dw2, db2 = _sink(gw2, w2), _sink(gb2, dy[0])
        dense_wgrad(h, dy, dw2, db2, sh2)
        dh = dense_dgrad(dy, w2, sh2, 0, w2.shape[0], relu_src=h)     # relu' fused in the dgrad epilogue
        dw1, db1 = _sink(gw1, w1), _sink(gb1, dh[0])
        dense_wgrad(x, dh, dw1, db1, sh1)
        dx = dense_dgrad(dh, w1, sh1, 0, w1.shape[0])
        n = lambda g, d: None if g is not None else d
        return dx, n(gw1, dw1), n(gb1, db1), n(gw2, dw2), n(gb2, db2), None, None, None, None, None, None


class ConvStackFn(torch.autograd.Function):
    """conv -> relu -> ... -> conv (no activation after the last): CNNResNorm.call_convs +
    last_conv, model/layers.py:30-38.  x [B,T,C].  params = (w0,b0,w1,b1,...), sinks likewise.
    shadows (tuple of ops.Shadow or None per layer, TTSMI_BF16): the convs run as bf16 implicit GEMMs
    (forward: window over x with W^T; dgrad: window over dy with the flipped-tap layout `wd`, ReLU' of
    the producing layer fused; wgrad: wgrad_rows with the conv window) - fp32 activations throughout."""

    # ---- bf16 "plain GEMM" route ---------------------------------------------------------------------------------
    # Channels-last makes the window of frame t the contiguous run x[t-p : t+p+1, :], so a 'same' Conv1D over a layout
    # with p zero rows around every sequence ([B, T + 2p, C], bf16) is an ordinary GEMM whose A rows OVERLAP: lda = C,
    # K = k C, row j starts at buffer row j and its result is the output at buffer row j + p.  The rows computed across a
    # sequence boundary land exactly on the margin rows of the output buffer and are zeroed (forward) or killed by the
    # ReLU' mask, whose margins are zero (backward).  That puts these 100-GFLOP launches on the persistent LDS-DMA GEMM
    # (gemm_bf16_dma_kernel: 691 / 781 TF at the reference-default shapes against 361 / 423 TF for the 64 x 128-tile
    # window kernel on fp32 activations; the fp32 results are bit-identical) and the weight gradients - one bf16 row-major
    # wgrad per tap, on shifted views - on wgrad_dma_kernel.  Every buffer carries 2p spare zero rows at its end so that
    # the shifted views of the last tap stay inside it.
    @staticmethod
    def _plain_ok(x, n_layers, shadows, params):
        if not _CONV_PLAIN or not shadows or x.dtype != torch.float32:
            return False
        k0 = params[0].shape[0]
        for j in range(n_layers):
            k, cin, cout = params[2 * j].shape
            if shadows[j] is None or k != k0 or k % 2 == 0 or k < 3 or cin % 8 or cout % 8:
                return False
        return True

    _MARGIN_IDX = {}

    @staticmethod
    def _margin_rows(B, T, p, device, mode):
        """Row indices to zero in a [B (T + 2p) + 2p, C] buffer: 'all' = every margin row and the spare tail;
        'ends' = the rows no GEMM row writes (the first p and everything from B (T + 2p) - p on).  Cached per shape:
        zeroing is then ONE index_fill launch instead of three slice fills."""
        key = (B, T, p, str(device), mode)
        idx = ConvStackFn._MARGIN_IDX.get(key)
        if idx is None:
            rows = B * (T + 2 * p)
            if mode == 'all':
                r = torch.arange(rows + 2 * p)
                t = r % (T + 2 * p)
                idx = r[(t < p) | (t >= T + p) | (r >= rows)]
            else:
                idx = torch.cat([torch.arange(p), torch.arange(rows - p, rows + 2 * p)])
            if len(ConvStackFn._MARGIN_IDX) > 64:
                ConvStackFn._MARGIN_IDX.clear()
            idx = ConvStackFn._MARGIN_IDX[key] = idx.to(device)
        return idx

    @staticmethod
    def _padded(B, T, p, C, dtype, device):
        """[B (T + 2p) + 2p, C] with zero margins / tail; returns (flat, interior view [B, T, C])."""
        rows = B * (T + 2 * p)
        flat = torch.empty((rows + 2 * p, C), dtype=dtype, device=device)
        flat.index_fill_(0, ConvStackFn._margin_rows(B, T, p, device, 'all'), 0)
        return flat, flat[:rows].view(B, T + 2 * p, C)[:, p:p + T]

    @staticmethod
    def _forward_plain(ctx, x, n_layers, shadows, params, sinks):
        B, T, _ = x.shape
        k = params[0].shape[0]
        p = (k - 1) // 2
        Mp = B * (T + 2 * p) - 2 * p                     # GEMM rows: every buffer row that has k rows below it
        cur, inner = ConvStackFn._padded(B, T, p, x.shape[2], torch.bfloat16, x.device)
        inner.copy_(x)                                   # fp32 -> bf16 into the padded layout, one launch
        saved = [cur]
        for j in range(n_layers):
            w, b = params[2 * j], params[2 * j + 1]
            _, cin, cout = w.shape
            last = j == n_layers - 1
            a_view = torch.as_strided(cur, (Mp, k * cin), (cin, 1))
            out = torch.empty((cur.shape[0], cout), dtype=torch.float32 if last else torch.bfloat16, device=x.device)
            hgemm_tn(a_view, shadows[j].wt, b, relu=not last, out=out[p:p + Mp])
            if last:
                y = out[:B * (T + 2 * p)].view(B, T + 2 * p, cout)[:, p:p + T].contiguous()
            else:
                out.index_fill_(0, ConvStackFn._margin_rows(B, T, p, x.device, 'all'), 0)
                saved.append(out)
                cur = out
        ctx.n = n_layers
        ctx.plain = (B, T, k, p, Mp)
        ctx.save_for_backward(*saved, *params[0::2])
        ctx.sinks = sinks
        ctx.shadows = shadows
        return y

    @staticmethod
    def _backward_plain(ctx, dy):
        n = ctx.n
        B, T, k, p, Mp = ctx.plain
        acts, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        sinks = ctx.sinks
        rows = B * (T + 2 * p)
        cout_last = ws[n - 1].shape[2]
        g, g_inner = ConvStackFn._padded(B, T, p, cout_last, torch.bfloat16, dy.device)
        g_inner.copy_(_c(dy))
        outs = [None] * (2 * n)
        dx = None
        for j in reversed(range(n)):
            gw, gb = (sinks[2 * j], sinks[2 * j + 1]) if sinks else (None, None)
            w = ws[j]
            _, cin, cout = w.shape
            dw, db = _sink(gw, w), _sink(gb, w[0, 0])
            xp = acts[j]
            # dW[tap] = x[. + tap]^T . g[. + p] over all buffer rows (the margins of g are zero, so are those of x)
            gv = g[p:p + rows]
            if _CONV_WGRAD_TAPS and rows % 32 == 0 and cin % 128 == 0 and cout % 128 == 0:
                # every tap in ONE launch (shifted-rows taps of ttsmi_hgemm_wgrad_rows): a third of the launches and slab
                # reductions; the buffers' 2p spare rows keep the last tap's reads inside them
                wgrad_rows_async(xp[:rows], gv, dw.reshape(k * cin, cout), db, conv=(k, 0, cin, 0))
            else:
                for tap in range(k):
                    wgrad_rows_async(xp[tap:tap + rows], gv, dw[tap], db if tap == 0 else None)
            if j > 0 or ctx.needs_input_grad[0]:
                a_view = torch.as_strided(g, (Mp, k * cout), (cout, 1))
                nxt = torch.empty((rows + 2 * p, cin), dtype=torch.bfloat16 if j > 0 else torch.float32, device=dy.device)
                hgemm_tn(a_view, ctx.shadows[j].wd, None, relu_src=xp[p:p + Mp] if j > 0 else None, out=nxt[p:p + Mp])
                if j > 0:                                 # rows no GEMM row writes; the mask zeroes the other margins
                    nxt.index_fill_(0, ConvStackFn._margin_rows(B, T, p, dy.device, 'ends'), 0)
                    g = nxt
                else:
                    inner = nxt[:rows].view(B, T + 2 * p, cin)[:, p:p + T]
                    sink = ctx.x_sink
                    if sink is not None and sink.buf is not None:       # (GradSink: one strided add instead of copy + add)
                        sink.buf.view(B, T, cin).add_(inner)
                    else:
                        dx = inner.contiguous()
            outs[2 * j] = None if gw is not None else dw
            outs[2 * j + 1] = None if gb is not None else db
        return (dx, None, None, *outs, *([None] * ctx.n_extra))

    @staticmethod
    def forward(ctx, x, n_layers, shadows, *args):
        """args = the 2 n parameters, then (optionally) their 2 n gradient sinks, then (optionally) a GradSink for x."""
        x = _c(x)
        params, sinks = args[:2 * n_layers], args[2 * n_layers:4 * n_layers]
        ctx.x_sink = args[4 * n_layers] if len(args) > 4 * n_layers else None
        ctx.n_extra = len(args) - 2 * n_layers
        ctx.plain = None
        if ConvStackFn._plain_ok(x, n_layers, shadows, params):
            return ConvStackFn._forward_plain(ctx, x, n_layers, shadows, params, sinks)
        B, T, _ = x.shape
        acts = [x]
        h = x
        use_h = []
        for j in range(n_layers):
            w, b = params[2 * j], params[2 * j + 1]
            k, cin, cout = w.shape
            sh = shadows[j] if shadows else None
            ok = sh is not None and cin % 8 == 0
            use_h.append(ok)
            if ok:
                h = hgemm_tn(h.reshape(B * T, cin), sh.wt, b, relu=(j < n_layers - 1),
                             conv=(k, T, cin, (k - 1) // 2)).reshape(B, T, cout)
            else:
                h = conv1d_fwd(h, w, b, relu=(j < n_layers - 1))
            acts.append(h)
        ctx.n = n_layers
        ctx.save_for_backward(*acts[:-1], *params[0::2])
        ctx.sinks = sinks
        ctx.shadows = shadows
        ctx.use_h = use_h
        return h

    @staticmethod
    def backward(ctx, dy):
        if ctx.plain is not None:
            return ConvStackFn._backward_plain(ctx, dy)
        n = ctx.n
        acts, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        sinks = ctx.sinks
        g = _c(dy)
        outs = [None] * (2 * n)
        for j in reversed(range(n)):
            gw, gb = (sinks[2 * j], sinks[2 * j + 1]) if sinks else (None, None)
            w = ws[j]
            k, cin, cout = w.shape
            B, T, _ = acts[j].shape
            dw, db = _sink(gw, w), _sink(gb, g[0, 0])
            relu_src = acts[j] if j > 0 else None
            if ctx.use_h[j]:
                sh = ctx.shadows[j]
                x2, g2 = acts[j].reshape(B * T, cin), g.reshape(B * T, cout)
                if cin % 128 == 0 and cout % 4 == 0:
                    wgrad_rows_async(x2, g2, dw.reshape(k * cin, cout), db, conv=(k, T, cin, (k - 1) // 2))
                else:
                    xT = cast_transpose_bf16(x2, taps=k, T=T, pad=(k - 1) // 2)
                    hgemm_wgrad(xT, cast_transpose_bf16(g2), dw.reshape(k * cin, cout), db, B * T)
                cp = _pad8(cout)
                if cp != cout:
                    g2 = torch.nn.functional.pad(g2, (0, cp - cout))
                g = hgemm_tn(g2, sh.wd, relu_src=None if relu_src is None else relu_src.reshape(B * T, cin),
                             conv=(k, T, cp, k - 1 - (k - 1) // 2)).reshape(B, T, cin)
            else:
                conv1d_wgrad(acts[j], g, dw, db)
                g = conv1d_dgrad(g, w, relu_src=relu_src)
            outs[2 * j] = None if gw is not None else dw
            outs[2 * j + 1] = None if gb is not None else db
        sink = ctx.x_sink
        if sink is not None and sink.buf is not None:
            sink.buf.view_as(g).add_(g)
            g = None
        return (g, None, None, *outs, *([None] * ctx.n_extra))


class AddLayerNormFn(torch.autograd.Function):
    """y = rowmask(keep_out(LN(keep_in(x) + res)*gamma + beta + s*PE)) - see include/ttsmi.h."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, ggamma, gbeta, pe, pe_scale, gpe_scale, T, row_pad,
                p_in, site_in, p_out, site_out, drop, relu_in, want_h=False, res_sink=None):
        ctx.stream_h = _stream()         # backward launches on the stream forward ran on (predictor side stream)
        ctx.res_sink = res_sink
        x = _c(x)
        res = None if res is None else _c(res)
        shp = x.shape
        C = shp[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        y_h = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_h else None
        mean = torch.empty((M,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
        seed = drop.seed if drop is not None else 0
        step_dev = drop.step_dev if drop is not None else None
        check(_lib.lib().ttsmi_add_layernorm_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(pe), _p(pe_scale),
                                                 int(T), _p(row_pad), float(p_in), int(site_in), float(p_out),
                                                 int(site_out), seed, _p(step_dev), LN_EPS, _p(y), _p(mean),
                                                 _p(rstd), M, C, _p(y_h), _stream()), 'add_layernorm_fwd')
        ctx.save_for_backward(x, res, gamma, mean, rstd, pe, pe_scale, row_pad, step_dev)
        ctx.cfg = (int(T), float(p_in), int(site_in), float(p_out), int(site_out), seed, bool(relu_in), M, C)
        ctx.sinks = (ggamma, gbeta, gpe_scale)
        if want_h:                      # the bf16 copy the first GEMM reads, written by the same kernel
            ctx.mark_non_differentiable(y_h)
            return y, y_h
        return y

    @staticmethod
    def backward(ctx, dy, _dy_h=None):
        with pin_stream(ctx.stream_h):
            return AddLayerNormFn._backward(ctx, dy) + (None, None)

    @staticmethod
    def _backward(ctx, dy):
        x, res, gamma, mean, rstd, pe, pe_scale, row_pad, step_dev = ctx.saved_tensors
        T, p_in, site_in, p_out, site_out, seed, relu_in, M, C = ctx.cfg
        ggamma, gbeta, gpe = ctx.sinks
        dy = _c(dy)
        dx = torch.empty_like(x)
        sink = getattr(ctx, 'res_sink', None)
        if res is None:
            dres = None
        elif p_in > 0 or relu_in or sink is not None:      # (a GradSink's buffer is added into: it cannot be dx)
            dres = torch.empty_like(x)
        else:
            dres = dx
        dgamma, dbeta = _sink(ggamma, gamma), _sink(gbeta, gamma)
        dps = None
        if pe is not None:
            dps = gpe if gpe is not None else torch.empty((1,), dtype=torch.float32, device=x.device)
        l = _lib.lib()
        ws = _ws(l.ttsmi_add_layernorm_bwd_ws_bytes(M, C), x.device)
        # deferred only when every parameter gradient lands in a caller-owned sink (autograd never sees it)
        defer = (_LN_PENDING is not None and ggamma is not None and gbeta is not None
                 and (pe is None or gpe is not None))
        check(l.ttsmi_add_layernorm_bwd(_p(dy), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), _p(pe),
                                        _p(pe_scale), T, _p(row_pad), p_in, site_in, p_out, site_out, seed,
                                        _p(step_dev), int(relu_in), _p(dx), _p(dres),
                                        None if defer else _p(dgamma), None if defer else _p(dbeta),
                                        None if defer else _p(dps), M, C, _p(ws), ws.numel(), None, _stream()),
              'add_layernorm_bwd')
        if defer:
            _ln_defer(ws, dgamma, dbeta, dps, M, C)
        if sink is not None and dres is not None:
            sink.buf = dres
        n = lambda g, d: None if g is not None else d
        dps_out = None
        if pe is not None and gpe is None:
            dps_out = dps.reshape(pe_scale.shape)
        return (dx, dres, n(ggamma, dgamma), n(gbeta, dbeta), None, None, None, dps_out, None, None, None,
                None, None, None, None, None, None)


def add_layernorm(x, res, gamma, beta, ggamma=None, gbeta=None, pe=None, pe_scale=None, gpe_scale=None,
                  T=0, row_pad=None, p_in=0.0, site_in=0, p_out=0.0, site_out=0, drop=None, relu_in=False, want_h=False,
                  res_sink=None):
    """want_h: returns (y, y as bf16) - the copy costs no launch of its own.  res_sink: see GradSink."""
    return AddLayerNormFn.apply(x, res, gamma, beta, ggamma, gbeta, pe, pe_scale, gpe_scale, T, row_pad,
                                p_in, site_in, p_out, site_out, drop, relu_in, want_h, res_sink)


class AttentionFn(torch.autograd.Function):
    """ctx = softmax(q k^T / sqrt(dh) + pad*-1e9) v per head, on the fused qkv projection output
    (model/layers.py:123-129,176-195,144-147).  Returns (ctx [M, H*dh], lse [B,H,T])."""

    @staticmethod
    def forward(ctx, qkv, key_pad, klen, B, H, T, dh, p_drop, drop, site, dtype=TTSMI_F32, dmask=None):
        """dmask: the layer's keep-bit table (attention_dropmask) for the bf16 kernels on fp32 tensors (TTSMI_BF16) - the
        same decisions as the in-kernel hash, read as bits by the forward and both backward kernels."""
        qkv = _c(qkv)
        d = H * dh
        if qkv.dtype == torch.bfloat16:        # bf16 tensors in and out (the same kernels the planned dense blocks launch)
            assert dh in (32, 64, 192), 'bf16 attention kernels are built for head dims 32 / 64 / 192'
            dtype = _lib.TTSMI_BF16_IO
        elif dtype != TTSMI_F32 and dh not in (32, 64, 192):
            dtype = TTSMI_F32                  # bf16 kernels are built for head dims 32 / 64 / 192
        if dtype not in (_lib.TTSMI_BF16, _lib.TTSMI_BF16_IO) or not p_drop > 0:
            dmask = None
        out = torch.empty((B * T, d), dtype=qkv.dtype, device=qkv.device)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
        seed = drop.seed if drop is not None else 0
        step_dev = drop.step_dev if drop is not None else None
        if dmask is not None:
            check(_lib.lib().ttsmi_attention_fwd_masked(_p(qkv), _p(key_pad), _p(klen), _p(out), _p(lse), B, H, T, dh,
                                                        float(p_drop), _p(dmask), int(dtype), _stream()),
                  'attention_fwd_masked')
        else:
            check(_lib.lib().ttsmi_attention_fwd(_p(qkv), _p(key_pad), _p(klen), _p(out), _p(lse), B, H, T, dh,
                                                 float(p_drop), seed, _p(step_dev), int(site), int(dtype),
                                                 _stream()), 'attention_fwd')
        ctx.save_for_backward(qkv, key_pad, klen, out, lse, step_dev)
        ctx.dmask = dmask
        ctx.cfg = (B, H, T, dh, float(p_drop), seed, int(site), int(dtype))
        ctx.mark_non_differentiable(lse)
        ctx.set_materialize_grads(False)      # no zero tensor for the unused d(lse)
        return out, lse

    @staticmethod
    def backward(ctx, dout, _dlse):
        qkv, key_pad, klen, out, lse, step_dev = ctx.saved_tensors
        B, H, T, dh, p_drop, seed, site, dtype = ctx.cfg
        dout = _c(dout)
        dqkv = torch.empty_like(qkv)
        l = _lib.lib()
        ws = _ws(l.ttsmi_attention_bwd_ws_bytes(B, H, T, dh), qkv.device)
        if ctx.dmask is not None:
            check(l.ttsmi_attention_bwd_masked(_p(qkv), _p(key_pad), _p(klen), _p(out), _p(dout), _p(lse), _p(dqkv),
                                               B, H, T, dh, p_drop, _p(ctx.dmask), _p(ws), ws.numel(), dtype, _stream()),
                  'attention_bwd_masked')
        else:
            check(l.ttsmi_attention_bwd(_p(qkv), _p(key_pad), _p(klen), _p(out), _p(dout), _p(lse), _p(dqkv),
                                        B, H, T, dh, p_drop, seed, _p(step_dev), site, _p(ws), ws.numel(),
                                        dtype, _stream()), 'attention_bwd')
        return dqkv, None, None, None, None, None, None, None, None, None, None, None


class EmbeddingFn(torch.autograd.Function):
    """models.py:522."""

    @staticmethod
    def forward(ctx, tokens, table, gtable):
        tokens = _c(tokens)
        M = tokens.numel()
        V, C = table.shape
        y = torch.empty((*tokens.shape, C), dtype=torch.float32, device=table.device)
        check(_lib.lib().ttsmi_embedding_fwd(_p(tokens), _p(table), _p(y), M, V, C, _stream()), 'embedding_fwd')
        ctx.save_for_backward(tokens)
        ctx.shape = (M, V, C)
        ctx.sink = gtable
        return y

    @staticmethod
    def backward(ctx, dy):
        tokens, = ctx.saved_tensors
        M, V, C = ctx.shape
        dy = _c(dy)
        dt = ctx.sink if ctx.sink is not None else torch.empty((V, C), dtype=torch.float32, device=dy.device)
        check(_lib.lib().ttsmi_embedding_bwd(_p(tokens), _p(dy), _p(dt), M, V, C, _stream()), 'embedding_bwd')
        return None, (None if ctx.sink is not None else dt), None


class PitchEmbedFn(torch.autograd.Function):
    """y = x + relu(p*w + b)   (models.py:527-531).  p [M] (flattened [B,Tp,1])."""

    @staticmethod
    def forward(ctx, x, p, w, b, gw, gb):
        x, p = _c(x), _c(p)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        check(_lib.lib().ttsmi_pitch_embed_fwd(_p(x), _p(p), _p(w), _p(b), _p(y), M, C, _stream()),
              'pitch_embed_fwd')
        ctx.save_for_backward(p, w, b)
        ctx.sinks = (gw, gb)
        ctx.mc = (M, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, w, b = ctx.saved_tensors
        gw, gb = ctx.sinks
        M, C = ctx.mc
        dy = _c(dy)
        dw, db = _sink(gw, w), _sink(gb, b)
        dp = torch.empty_like(p) if ctx.needs_input_grad[1] else None
        l = _lib.lib()
        ws = _ws(l.ttsmi_pitch_embed_bwd_ws_bytes(M, C), dy.device)
        check(l.ttsmi_pitch_embed_bwd(_p(dy), _p(p), _p(w), _p(b), _p(dp), _p(dw), _p(db), M, C, _p(ws),
                                      ws.numel(), _stream()), 'pitch_embed_bwd')
        return dy, dp, (None if gw is not None else dw), (None if gb is not None else db), None, None


class RowDotFn(torch.autograd.Function):
    """y[m] = act(x[m,:].w + b) * (1 - row_pad[m])   (layers.py:479,484-485)."""

    @staticmethod
    def forward(ctx, x, w, b, gw, gb, row_pad, relu):
        ctx.stream_h = _stream()         # backward launches on the stream forward ran on (predictor side stream)
        x = _c(x)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty((*x.shape[:-1], 1), dtype=torch.float32, device=x.device)
        check(_lib.lib().ttsmi_rowdot_fwd(_p(x), _p(w), _p(b), _p(row_pad), _p(y), M, C, int(relu),
                                          _stream()), 'rowdot_fwd')
        ctx.save_for_backward(x, w, y, row_pad)
        ctx.cfg = (M, C, bool(relu))
        ctx.sinks = (gw, gb)
        return y

    @staticmethod
    def backward(ctx, dy):
        with pin_stream(ctx.stream_h):
            return RowDotFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, w, y, row_pad = ctx.saved_tensors
        M, C, relu = ctx.cfg
        gw, gb = ctx.sinks
        dy = _c(dy)
        dx = torch.empty_like(x)
        dw = _sink(gw, w)
        db = gb if gb is not None else torch.empty((1,), dtype=torch.float32, device=x.device)
        l = _lib.lib()
        ws = _ws(l.ttsmi_rowdot_bwd_ws_bytes(M, C), x.device)
        check(l.ttsmi_rowdot_bwd(_p(dy), _p(y), _p(x), _p(w), _p(row_pad), _p(dx), _p(dw), _p(db), M, C,
                                 int(relu), _p(ws), ws.numel(), _stream()), 'rowdot_bwd')
        return dx, (None if gw is not None else dw), (None if gb is not None else db), None, None, None, None


class RowMaskFn(torch.autograd.Function):
    """y = x * (1 - row_pad)   (layers.py:482)."""

    @staticmethod
    def forward(ctx, x, row_pad):
        ctx.stream_h = _stream()         # backward launches on the stream forward ran on (predictor side stream)
        x = _c(x)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        check(_lib.lib().ttsmi_rowmask_mul(_p(x), _p(row_pad), _p(y), M, C, _stream()), 'rowmask_mul')
        ctx.save_for_backward(row_pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        with pin_stream(ctx.stream_h):
            return RowMaskFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        row_pad, = ctx.saved_tensors
        dy = _c(dy)
        C = dy.shape[-1]
        M = dy.numel() // C
        dx = torch.empty_like(dy)
        check(_lib.lib().ttsmi_rowmask_mul(_p(dy), _p(row_pad), _p(dx), M, C, _stream()), 'rowmask_mul')
        return dx, None


class BranchFn(torch.autograd.Function):
    """Identity that gives a side-stream branch ONE gradient edge back to the tensor it forks from.  Called inside the
    side stream's context, its backward node belongs to that stream, so the gradients of the branch's several consumers
    (the two StatPredictors) are summed there; without it autograd sums them on the stream of the tensor's PRODUCER'S
    consumer - the main stream, which then has to wait for the whole side chain at the moment the second gradient
    arrives (before the decoder's backward has been issued when the branch's backward is replayed first)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def wgrad_stream():
    """The side stream the weight gradients run on (None until the first overlapped launch)."""
    W = _WgradStream.cur()
    return W.stream if W.pending else None


class LenRegFn(torch.autograd.Function):
    """Expand (model/layers.py:549-565): y[b,j] = x[b, idx[b,j]] (0 where idx < 0)."""

    @staticmethod
    def forward(ctx, x, idx, cum, hook=None):
        # hook (per model, set by dp.DataParallel): called at the start of backward, i.e. when backward crosses
        # from the decoder into the encoder - every decoder-side gradient has been launched by then, so data
        # parallel starts the all-reduce of the decoder half of the flat gradient buffer underneath the
        # encoder's backward.
        x = _c(x)
        B, Tp, C = x.shape
        cap = idx.shape[1]
        ctx.hook = hook
        y = torch.empty((B, cap, C), dtype=torch.float32, device=x.device)
        check(_lib.lib().ttsmi_lenreg_fwd(_p(x), _p(idx), _p(y), B, Tp, cap, C, _stream()), 'lenreg_fwd')
        ctx.save_for_backward(cum)
        ctx.shape = (B, Tp, cap, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        cum, = ctx.saved_tensors
        B, Tp, cap, C = ctx.shape
        if ctx.hook is not None:
            ln_flush()                      # the decoder's LayerNorm gradients must be final before their all-reduce
                                            # (each reduce runs on the stream that produced its partial sums; the hook's
                                            # all-reduce is ordered behind the main AND the weight-gradient stream)
            ctx.hook()
        dy = _c(dy)
        dx = torch.empty((B, Tp, C), dtype=torch.float32, device=dy.device)
        check(_lib.lib().ttsmi_lenreg_bwd(_p(dy), _p(cum), _p(dx), B, Tp, cap, C, _stream()), 'lenreg_bwd')
        return dx, None, None, None


class L1LossFn(torch.autograd.Function):
    """mean |target - pred| over every element (utils/losses.py:41-49 with mask=None; integer
    targets are cast to float).  sign(p-t)/n is produced in the same pass and kept for backward."""

    @staticmethod
    def forward(ctx, pred, target):
        assert pred.dim() >= 2 and pred.stride(-1) == 1
        cols = pred.shape[-1]
        rows = pred.numel() // cols
        p2 = pred.reshape(rows, cols) if pred.is_contiguous() else _c(pred).reshape(rows, cols)
        target = _c(target)
        assert target.numel() == rows * cols, (target.shape, pred.shape)
        assert target.dtype in (torch.float32, torch.int32)
        grad = torch.empty((rows, cols), dtype=torch.float32, device=pred.device)
        loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
        l = _lib.lib()
        ws = _ws(l.ttsmi_l1_loss_ws_bytes(rows * cols), pred.device)
        check(l.ttsmi_l1_loss(_p(p2), p2.stride(0), _p(target), int(target.dtype == torch.int32), rows, cols,
                              1.0, _p(grad), cols, _p(loss), _p(ws), ws.numel(), _stream()), 'l1_loss')
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dl):
        grad, = ctx.saved_tensors
        return (grad.reshape(ctx.shape) * dl), None


class WeightedL1LossesFn(torch.autograd.Function):
    """weighted_sum_losses over L1 terms (utils/losses.py:63-70) as ONE op: apply(coeffs, unit_seed, pred0, target0,
    pred1, target1, ...) -> (total, loss0, loss1, ...).  One launch per term plus one that finishes every mean and the
    weighted total (ttsmi_l1_losses_weighted); the terms' gradients coeff_i sign(p - t) / n_i are written by the same
    pass.  The three separate losses and the Python-level weighted sum were ~18 launches of scalar glue per step between
    the end of the forward and the start of the backward, where nothing else keeps the GPU busy.
    Only `total` is differentiable.  unit_seed=True promises that backward is seeded with 1 (loss.backward()): the saved
    gradients are then returned as they are instead of being multiplied by the incoming scalar."""

    @staticmethod
    def forward(ctx, coeffs, unit_seed, *pt):
        """coeffs: tuple of weights, optionally followed by ONE tuple of per-term divisors (global element counts of a
        batch sharded over data-parallel ranks; 0 = the term's own count): ((c0, c1, ...), (n0, n1, ...))."""
        denoms = None
        if len(coeffs) == 2 and isinstance(coeffs[0], (tuple, list)):
            coeffs, denoms = coeffs
        assert len(pt) % 2 == 0 and 2 <= len(pt) <= 16 and len(coeffs) == len(pt) // 2
        n = len(pt) // 2
        assert denoms is None or len(denoms) == n
        preds, targets, grads, shapes = [], [], [], []
        for i in range(n):
            pred, target = pt[2 * i], _c(pt[2 * i + 1])
            assert pred.dim() >= 2 and pred.stride(-1) == 1
            cols = pred.shape[-1]
            rows = pred.numel() // cols
            p2 = pred.reshape(rows, cols) if pred.is_contiguous() else _c(pred).reshape(rows, cols)
            assert target.numel() == rows * cols, (target.shape, pred.shape)
            assert target.dtype in (torch.float32, torch.int32)
            preds.append(p2)
            targets.append(target)
            grads.append(torch.empty((rows, cols), dtype=torch.float32, device=pred.device))
            shapes.append(pred.shape)
        dev = preds[0].device
        out = torch.empty((n + 1,), dtype=torch.float32, device=dev)          # [loss_0 .. loss_{n-1}, total]
        l = _lib.lib()
        ws = _ws(l.ttsmi_l1_losses_weighted_ws_bytes(n), dev)
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        i64 = lambda vs: (ctypes.c_int64 * n)(*[int(v) for v in vs])
        check(l.ttsmi_l1_losses_weighted(
            n, ptrs(preds), i64(t.stride(0) for t in preds), ptrs(targets),
            (ctypes.c_int32 * n)(*[int(t.dtype == torch.int32) for t in targets]), i64(t.shape[0] for t in preds),
            i64(t.shape[1] for t in preds), (ctypes.c_float * n)(*[float(c) for c in coeffs]),
            None if denoms is None else i64(denoms), ptrs(grads),
            i64(t.stride(0) for t in grads), _p(out), out.data_ptr() + 4 * n, _p(ws), ws.numel(), _stream()),
            'l1_losses_weighted')
        ctx.save_for_backward(*grads)
        ctx.shapes, ctx.unit_seed = shapes, bool(unit_seed)
        losses = [out[i] for i in range(n)]
        ctx.mark_non_differentiable(*losses)
        ctx.set_materialize_grads(False)          # no zero tensors (a fill launch each) for the terms' unused gradients
        return (out[n], *losses)

    @staticmethod
    def backward(ctx, dtotal, *_):
        if dtotal is None:
            return (None, None) + (None,) * (2 * len(ctx.shapes))
        res = [None, None]
        for g, shape in zip(ctx.saved_tensors, ctx.shapes):
            g = g.reshape(shape)
            res += [g if ctx.unit_seed else g * dtotal, None]
        return tuple(res)


class ConvReluPreMaskedFn(torch.autograd.Function):
    """h = relu(conv1d(x)).  INTERNAL to the predictor layer (layers.py:512-513): its backward
    expects dy ALREADY multiplied by (h > 0) - the following AddLayerNormFn(relu_in=True) does that
    inside its own backward kernel, so no separate relu' pass exists."""

    @staticmethod
    def forward(ctx, x, w, b, gw, gb, sh=None):
        ctx.stream_h = _stream()         # backward launches on the stream forward ran on (predictor side stream)
        x = _c(x)
        B, T, Cin = x.shape
        k, _, Cout = w.shape
        if sh is not None and Cin % 8 == 0:
            h = hgemm_tn(x.reshape(B * T, Cin), sh.wt, b, True, conv=(k, T, Cin, (k - 1) // 2)).reshape(B, T, Cout)
        else:
            sh = None
            h = conv1d_fwd(x, w, b, relu=True)
        ctx.save_for_backward(x, w)
        ctx.sinks = (gw, gb)
        ctx.sh = sh
        return h

    @staticmethod
    def backward(ctx, dh):
        with pin_stream(ctx.stream_h):
            return ConvReluPreMaskedFn._backward(ctx, dh)

    @staticmethod
    def _backward(ctx, dh):
        x, w = ctx.saved_tensors
        gw, gb = ctx.sinks
        dh = _c(dh)
        dw, db = _sink(gw, w), _sink(gb, dh[0, 0])
        sh = ctx.sh
        B, T, Cin = x.shape
        k, _, Cout = w.shape
        if sh is not None and Cin % 128 == 0 and Cout % 4 == 0:
            hgemm_wgrad_rows(x.reshape(B * T, Cin), dh.reshape(B * T, Cout), dw.reshape(k * Cin, Cout), db,
                             conv=(k, T, Cin, (k - 1) // 2))
        elif sh is not None:
            xT = cast_transpose_bf16(x.reshape(B * T, Cin), taps=k, T=T, pad=(k - 1) // 2)
            hgemm_wgrad(xT, cast_transpose_bf16(dh.reshape(B * T, Cout)), dw.reshape(k * Cin, Cout), db, B * T)
        else:
            conv1d_wgrad(x, dh, dw, db)
        dx = None
        if ctx.needs_input_grad[0]:
            if sh is not None:
                dh2, Cp = dh.reshape(B * T, Cout), _pad8(Cout)
                if Cp != Cout:             # filters 226: zero-pad dy to 232 columns, wd is laid out to match
                    dh2 = torch.nn.functional.pad(dh2, (0, Cp - Cout))
                dx = hgemm_tn(dh2, sh.wd, conv=(k, T, Cp, k - 1 - (k - 1) // 2)).reshape(B, T, Cin)
            else:
                dx = conv1d_dgrad(dh, w)
        return dx, (None if gw is not None else dw), (None if gb is not None else db), None, None, None


# =================================================================================================
# One autograd node per StatPredictor (model/layers.py:481-485, 510-524)
# =================================================================================================
class _SubCtx:
    """What our own Functions use of an autograd ctx, for a Function that runs INSIDE another Function's forward /
    backward (torch.autograd.Function.apply costs ~20 us of host time per node and direction - tools/debug/launch_cost.py -
    and a StatPredictor was eight nodes around ~25 launches; the activation gradient is the only one that travels
    between them: every parameter gradient goes to its caller-owned sink)."""

    def __init__(self, n_inputs):
        self.needs_input_grad = (True,) + (False,) * (n_inputs - 1)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class StatPredictorFn(torch.autograd.Function):
    """RowMask -> [conv + relu -> LayerNorm (+ dropout)] x n -> row dot, as ONE autograd node: the member Functions' own
    forward / _backward bodies run back to back (same launches, same order, same stream as the separate nodes).
    convs: [(w, b, gw, gb, shadow)], lns: [(gamma, beta, ggamma, gbeta, site_out)], lin: (w, b, gw, gb)."""

    @staticmethod
    def forward(ctx, x, pad, convs, lns, lin, relu_head, rate, drop):
        ctx.stream_h = _stream()         # backward launches on the stream forward ran on (predictor side stream)
        subs = []
        c = _SubCtx(2)
        h = RowMaskFn.forward(c, x, pad)
        subs.append((RowMaskFn, c, None))
        for (w, b, gw, gb, sh), (gamma, beta, ggamma, gbeta, site) in zip(convs, lns):
            c = _SubCtx(6)
            h = ConvReluPreMaskedFn.forward(c, h, w, b, gw, gb, sh)
            subs.append((ConvReluPreMaskedFn, c, tuple(h.shape)))
            B, T, C = h.shape
            c = _SubCtx(18)
            h = AddLayerNormFn.forward(c, h.reshape(B * T, C), None, gamma, beta, ggamma, gbeta, None, None, None, 0, None,
                                       0.0, 0, rate, site, drop, True, False).reshape(B, T, C)
            subs.append((AddLayerNormFn, c, None))
        c = _SubCtx(7)
        y = RowDotFn.forward(c, h, lin[0], lin[1], lin[2], lin[3], pad, relu_head)
        subs.append((RowDotFn, c, None))
        ctx.subs = subs
        return y

    @staticmethod
    def backward(ctx, dy):
        with pin_stream(ctx.stream_h):
            g = dy
            for cls, c, shape in reversed(ctx.subs):
                if shape is not None:
                    g = g.reshape(shape)                 # the conv's output gradient [B, T, C] (its LayerNorm worked on rows)
                g = cls._backward(c, g)[0]
        ctx.subs = None
        return g, None, None, None, None, None, None, None


# =================================================================================================
# One autograd node per SelfAttentionDenseBlock (model/layers.py:214-230)
# =================================================================================================
def to_bf16(t):
    """bf16 copy of an fp32 tensor (one streaming kernel)."""
    t = _c(t)
    out = torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)
    check(_lib.lib().ttsmi_cast_f32_to_bf16(_p(t), _p(out), t.numel(), _stream()), 'cast_f32_to_bf16')
    return out


def _ln_fwd(x, res, gamma, beta, row_pad, p_in, site_in, drop, want_h=False):
    """Returns (y, y_bf16 or None, mean, rstd)."""
    M, C = x.shape
    y = torch.empty_like(x)
    y_h = torch.empty((M, C), dtype=torch.bfloat16, device=x.device) if want_h else None
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(_lib.lib().ttsmi_add_layernorm_fwd(_p(x), _p(res), _p(gamma), _p(beta), None, None, 0, _p(row_pad),
                                             float(p_in), int(site_in), 0.0, 0, drop.seed, _p(drop.step_dev),
                                             LN_EPS, _p(y), _p(mean), _p(rstd), M, C, _p(y_h), _stream()),
          'add_layernorm_fwd')
    return y, y_h, mean, rstd


def _ln_bwd(dy, x, res, gamma, mean, rstd, row_pad, p_in, site_in, drop, dgamma, dbeta, dx_bf16=False):
    """Returns (dx, dres); dres aliases dx when the x branch has no dropout.  dx_bf16: dx comes back
    as bf16 (it only feeds GEMMs); the fp32 dx is then written only where dres has to alias it."""
    M, C = x.shape
    if dx_bf16:
        dx_h = torch.empty((M, C), dtype=torch.bfloat16, device=x.device)
        dres = torch.empty_like(x)
        dx = None if p_in > 0 else dres
    else:
        dx_h = None
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if p_in > 0 else dx
    l = _lib.lib()
    ws = _ws(l.ttsmi_add_layernorm_bwd_ws_bytes(M, C), x.device)
    defer = _LN_PENDING is not None           # dgamma / dbeta are always gradient sinks of the flat buffer here
    check(l.ttsmi_add_layernorm_bwd(_p(dy), _p(x), _p(res), _p(gamma), _p(mean), _p(rstd), None, None, 0,
                                    _p(row_pad), float(p_in), int(site_in), 0.0, 0, drop.seed, _p(drop.step_dev),
                                    0, _p(dx), _p(dres), None if defer else _p(dgamma), None if defer else _p(dbeta),
                                    None, M, C, _p(ws), ws.numel(), _p(dx_h), _stream()), 'add_layernorm_bwd')
    if defer:
        _ln_defer(ws, dgamma, dbeta, None, M, C)
    return (dx_h if dx_bf16 else dx), dres


class DenseBlockFn(torch.autograd.Function):
    """SelfAttentionDenseBlock.call (model/layers.py:226-230) as ONE autograd node:
        qkv = h.Wqkv+b -> attention -> [h | ctx].Wo+b -> a = LN(drop(.) + h)*mask
        -> relu(a.W1+b1).W2+b2 -> out = LN(drop(.) + a)*mask
    with a hand-ordered backward.  Compared with chaining the per-layer Functions this (i) costs one
    Python autograd node instead of seven per block (the eager host loop was as long as the GPU
    step), (ii) lets the three gradient contributions of `h` and the two of `a` be summed inside the
    dgrad GEMM epilogues (accumulate) instead of by separate full-tensor add kernels, and (iii) frees
    each intermediate gradient as soon as its consumers are launched.
    P = dict of the block's parameter tensors, G = dict of their gradient sinks (views of the flat
    gradient buffer), S = dict of bf16 shadows (empty for the fp32 path)."""

    @staticmethod
    def forward(ctx, h, h_bf, P, G, S, pad, klen, B, H, T, rate, drop, sites, dtype, want_lse, dmask_pre=None):
        h = _c(h)
        M, d = h.shape
        dh_ = d // H
        F = P['ffn.w1'].shape[1]
        if dtype != TTSMI_F32 and dh_ not in (32, 64, 192):
            dtype = TTSMI_F32
        all_h = (dtype != TTSMI_F32 and all(S.get(k) is not None for k in ('wqkv', 'wo', 'ffn.w1', 'ffn.w2'))
                 and d % 64 == 0 and F % 8 == 0)
        if all_h:
            # TTSMI_BF16: every tensor that is only a GEMM / attention operand lives in HBM as bf16 (qkv, ctx,
            # the FFN hidden h1, and in backward df, do, dctx, dqkv, dh1); the residual stream (h, o, a, f, and
            # the gradients da, dh) stays fp32, and LayerNorm emits a bf16 copy of its output for the GEMMs.
            if h_bf is None:
                h_bf = to_bf16(h)
            qkv = hgemm_tn(h_bf, S['wqkv'].wt, P['bqkv'], out_bf16=True)
            dtype = _lib.TTSMI_BF16_IO
            cx = torch.empty((M, d), dtype=torch.bfloat16, device=h.device)
        else:
            h_bf = None
            qkv = dense_fwd(h, P['wqkv'], P['bqkv'], False, None, S.get('wqkv'))
            cx = torch.empty((M, d), dtype=torch.float32, device=h.device)
        lse = torch.empty((B, H, T), dtype=torch.float32, device=h.device)
        dmask = None
        if all_h and rate > 0 and _ATTN_DROPBITS:
            # dropout decisions of this layer as a bit table, evaluated once: forward and both backward kernels
            # read bits instead of hashing in their inner loops.  dmask_pre: the model generated the table ahead of
            # time on a side stream (ForwardTransformer._launch_dropmasks); otherwise it is generated here.
            dmask = dmask_pre if dmask_pre is not None else attention_dropmask(B, H, T, rate, drop, sites[0], h.device)
            check(_lib.lib().ttsmi_attention_fwd_masked(_p(qkv), _p(pad), _p(klen), _p(cx), _p(lse), B, H, T, dh_,
                                                        float(rate), _p(dmask), _lib.TTSMI_BF16_IO, _stream()),
                  'attention_fwd_masked')
        else:
            check(_lib.lib().ttsmi_attention_fwd(_p(qkv), _p(pad), _p(klen), _p(cx), _p(lse), B, H, T, dh_, float(rate),
                                                 drop.seed, _p(drop.step_dev), sites[0], int(dtype), _stream()),
                  'attention_fwd')
        if all_h:
            o = hgemm_tn(h_bf, S['wo'].wt, P['bo'], a2=cx)
            a, a_bf, mean1, rstd1 = _ln_fwd(o, h, P['ln1.gamma'], P['ln1.beta'], pad, rate, sites[1], drop, True)
            h1 = hgemm_tn(a_bf, S['ffn.w1'].wt, P['ffn.b1'], relu=True, out_bf16=True)
            f = hgemm_tn(h1, S['ffn.w2'].wt, P['ffn.b2'])
            out, out_bf, mean2, rstd2 = _ln_fwd(f, a, P['ln2.gamma'], P['ln2.beta'], pad, rate, sites[2], drop, True)
        else:
            o = dense_fwd(h, P['wo'], P['bo'], False, cx, S.get('wo'))
            a, a_bf, mean1, rstd1 = _ln_fwd(o, h, P['ln1.gamma'], P['ln1.beta'], pad, rate, sites[1], drop)
            h1 = dense_fwd(a, P['ffn.w1'], P['ffn.b1'], True, None, S.get('ffn.w1'))
            f = dense_fwd(h1, P['ffn.w2'], P['ffn.b2'], False, None, S.get('ffn.w2'))
            out, out_bf, mean2, rstd2 = _ln_fwd(f, a, P['ln2.gamma'], P['ln2.beta'], pad, rate, sites[2], drop)
        ctx.save_for_backward(h, h_bf, qkv, cx, lse, o, a, a_bf, h1, f, mean1, rstd1, mean2, rstd2, pad, klen, dmask)
        ctx.cfg = (P, G, S, B, H, T, dh_, float(rate), drop, sites, int(dtype), all_h)
        if out_bf is None:
            out_bf = out.new_empty(0)
        ctx.mark_non_differentiable(out_bf, qkv, lse)
        # without this autograd materialises ZERO gradients for the three auxiliary outputs before every
        # backward call: ~60 MB of fill kernels per decoder block
        ctx.set_materialize_grads(False)
        return out, out_bf, qkv, lse

    @staticmethod
    def backward(ctx, dout, _dout_bf, _dqkv, _dlse):
        h, h_bf, qkv, cx, lse, o, a, a_bf, h1, f, mean1, rstd1, mean2, rstd2, pad, klen, dmask = ctx.saved_tensors
        P, G, S, B, H, T, dh_, rate, drop, sites, dtype, all_h = ctx.cfg
        d = h.shape[1]
        dout = _c(dout)
        l = _lib.lib()
        if all_h:
            s1, s2, sho, shq = S['ffn.w1'], S['ffn.w2'], S['wo'], S['wqkv']
            # ---- LN2 + FFN ---------------------------------------------------------------------
            df, da = _ln_bwd(dout, f, a, P['ln2.gamma'], mean2, rstd2, pad, rate, sites[2], drop,
                             G['ln2.gamma'], G['ln2.beta'], dx_bf16=True)
            wgrad_rows_async(h1, df, G['ffn.w2'], G['ffn.b2'])
            dh1 = hgemm_tn(df, s2.wb, relu_src=h1, out_bf16=True)                            # relu' fused, bf16 out
            wgrad_rows_async(a_bf, dh1, G['ffn.w1'], G['ffn.b1'])
            hgemm_tn(dh1, s1.wb, out=da, accumulate=True)                                    # da += dh1.W1^T
            del dh1, df
            # ---- LN1 + output projection ---------------------------------------------------------
            do, dh = _ln_bwd(da, o, h, P['ln1.gamma'], mean1, rstd1, pad, rate, sites[1], drop,
                             G['ln1.gamma'], G['ln1.beta'], dx_bf16=True)
            del da
            wgrad_rows_async(h_bf, do, G['wo'][:d], G['bo'])
            wgrad_rows_async(cx, do, G['wo'][d:], None)
            hgemm_tn(do, sho.wb[:d], out=dh, accumulate=True)                                # dh += do.Wo_top^T
            dctx = hgemm_tn(do, sho.wb[d:2 * d], out_bf16=True)
            del do
            # ---- attention + qkv projection ------------------------------------------------------
            dqkv = torch.empty_like(qkv)
            ws = _ws(l.ttsmi_attention_bwd_ws_bytes(B, H, T, dh_), h.device)
            if dmask is not None:
                check(l.ttsmi_attention_bwd_masked(_p(qkv), _p(pad), _p(klen), _p(cx), _p(dctx), _p(lse), _p(dqkv), B, H,
                                                   T, dh_, rate, _p(dmask), _p(ws), ws.numel(), _lib.TTSMI_BF16_IO,
                                                   _stream()), 'attention_bwd_masked')
            else:
                check(l.ttsmi_attention_bwd(_p(qkv), _p(pad), _p(klen), _p(cx), _p(dctx), _p(lse), _p(dqkv), B, H, T, dh_,
                                            rate, drop.seed, _p(drop.step_dev), sites[0], _p(ws), ws.numel(), dtype,
                                            _stream()), 'attention_bwd')
            wgrad_rows_async(h_bf, dqkv, G['wqkv'], G['bqkv'])
            hgemm_tn(dqkv, shq.wb, out=dh, accumulate=True)                                  # dh += dqkv.Wqkv^T
            return (dh,) + (None,) * 15
        # ---- LN2 + FFN -----------------------------------------------------------------------
        df, da = _ln_bwd(dout, f, a, P['ln2.gamma'], mean2, rstd2, pad, rate, sites[2], drop,
                         G['ln2.gamma'], G['ln2.beta'])
        if da is df:                       # the GEMM below accumulates into da: it must own its buffer
            da = df.clone()
        dense_wgrad(h1, df, G['ffn.w2'], G['ffn.b2'], S.get('ffn.w2'))
        dh1 = dense_dgrad(df, P['ffn.w2'], S.get('ffn.w2'), 0, P['ffn.w2'].shape[0], relu_src=h1)
        dense_wgrad(a, dh1, G['ffn.w1'], G['ffn.b1'], S.get('ffn.w1'))
        dense_dgrad(dh1, P['ffn.w1'], S.get('ffn.w1'), 0, d, out=da, accumulate=True)      # da += dh1.W1^T
        del dh1
        # ---- LN1 + output projection -----------------------------------------------------------
        do, dh = _ln_bwd(da, o, h, P['ln1.gamma'], mean1, rstd1, pad, rate, sites[1], drop,
                         G['ln1.gamma'], G['ln1.beta'])
        if dh is do:
            dh = do.clone()
        del da
        sho = S.get('wo')
        dyT = dense_wgrad(h, do, G['wo'][:d], G['bo'], sho)
        dense_wgrad(cx, do, G['wo'][d:], None, sho, dyT)
        dense_dgrad(do, P['wo'], sho, 0, d, out=dh, accumulate=True)                        # dh += do.Wo_top^T
        dctx = dense_dgrad(do, P['wo'], sho, d, 2 * d)
        del do
        # ---- attention + qkv projection --------------------------------------------------------
        dqkv = torch.empty_like(qkv)
        ws = _ws(l.ttsmi_attention_bwd_ws_bytes(B, H, T, dh_), h.device)
        check(l.ttsmi_attention_bwd(_p(qkv), _p(pad), _p(klen), _p(cx), _p(dctx), _p(lse), _p(dqkv), B, H, T, dh_,
                                    rate, drop.seed, _p(drop.step_dev), sites[0], _p(ws), ws.numel(), dtype,
                                    _stream()), 'attention_bwd')
        shq = S.get('wqkv')
        dense_wgrad(h, dqkv, G['wqkv'], G['bqkv'], shq)
        dense_dgrad(dqkv, P['wqkv'], shq, 0, d, out=dh, accumulate=True)                    # dh += dqkv.Wqkv^T
        return (dh,) + (None,) * 15


# =================================================================================================
# The same block with its launch sequence issued from C++ (ttsmi_dense_block_fwd / _bwd, include/ttsmi.h)
# =================================================================================================
# stream priority of the weight-gradient side stream (torch: -1 = high, 0 = default)
_WGRAD_PRIO = int(os.environ.get('TTSMI_WGRAD_PRIO', '0'))
_CONV_WGRAD_TAPS = os.environ.get('TTSMI_CONV_WGRAD_TAPS', '1') != '0'     # A/B knob: 0 = one weight-gradient launch per conv tap
FUSE_LN_MIN_ROWS_INFERENCE = int(os.environ.get('TTSMI_FUSE_LN_MIN_ROWS', '8192'))


_CHAIN_BWD = os.environ.get('TTSMI_DENSE_CHAIN_BWD', '1') != '0'
# rows from which the chain kernels replace the four / three launches.  Round 5: 16 384 (the 128-row form's workgroup lives ~60 us
# whatever the row count).  Round 6: the 64-row form (four waves, csrc/chain16.h) wins alone from ~2 000 rows on - 52 against 56 us
# forward and 37 against 40 us backward at 6 400 rows, 56 / 67 and 40 / 49 at 12 000 (profiles/r06_chain64_alone.txt) - and in
# the step saves the launches' boundaries as well
CHAIN_MIN_ROWS = int(os.environ.get('TTSMI_DENSE_CHAIN_MIN_ROWS', '1024'))


class DenseBlockPlan:
    """Persistent buffers + the filled `ttsmi_dense_block` descriptor of ONE dense block, sized for a CAPACITY of rows.

    The per-launch Python path (DenseBlockFn) spends ~14 us of host time per kernel launch (allocator, ctypes
    argument conversion, autograd bookkeeping); at ~400 launches a step that is 6 ms of enqueueing for 5 ms of GPU
    work.  A plan owns every activation / temporary of its block, so the descriptor is filled once and a training
    step costs two ctypes calls per block (forward, backward) whose launches are issued from C++.
    `shared` holds the buffers that blocks of one stack may share (nothing outside the main stream reads them).

    Every buffer is a row-major [rows, C] matrix (or a flat per-row vector), so a batch with fewer rows than the
    capacity uses a prefix of it: `rebind(B, T)` re-targets the plan at a new batch shape without allocating - with
    length-bucketed training data almost every batch has its own (B, Tp, Tm), and a plan per exact shape meant a
    device synchronisation plus ~30 allocations per block on almost every step (round-2 advisor finding)."""

    def __init__(self, P, G, S, B, H, T, device, shared, fuse_ln=True, backward=True, cap_rows=0, chain=False):
        """backward=False: a forward-only plan (inference) - the backward temporaries are not allocated.
        chain: the forward's row-local chain (o-projection + res-norm 1 -> FFN -> res-norm 2 -> the next block's qkv
        projection) as ONE launch (csrc/chain.hip, ttsmi_dense_block.chain_w) - training plans with fused LayerNorms and a
        bf16 residual stream only."""
        l = _lib.lib()
        self.backward = bool(backward)
        self.pack_ev = None                       # event of a weight-stream pack issued ahead of this step on a side stream
        d = P['wqkv'].shape[0]
        F = P['ffn.w1'].shape[1]
        cap = self.cap = max(int(cap_rows), B * T)
        bf, f32 = torch.bfloat16, torch.float32
        e = lambda shape, dt: torch.empty(shape, dtype=dt, device=device)
        self.H, self.d, self.F = H, d, F
        self.t = t = {}
        self.want_fuse = bool(fuse_ln) and d == 256
        for name, shape, dt in (('qkv', (cap, 3 * d), bf), ('cx', (cap, d), bf), ('a_bf', (cap, d), bf), ('h1', (cap, F), bf),
                                ('out_bf', (cap, d), bf), ('lse', (cap * H,), f32), ('o', (cap, d), f32), ('a', (cap, d), f32),
                                ('f', (cap, d), f32), ('out', (cap, d), f32), ('mean1', (cap,), f32), ('rstd1', (cap,), f32),
                                ('mean2', (cap,), f32), ('rstd2', (cap,), f32),
                                # read by the weight-gradient stream: private to the block
                                ('df', (cap, d), bf), ('dh1', (cap, F), bf), ('d_o', (cap, d), bf), ('dqkv', (cap, 3 * d), bf),
                                ('dh', (cap, d), f32)):
            unused = (not backward and name in ('df', 'dh1', 'd_o', 'dqkv', 'dh')) or (self.want_fuse and backward and
                                                                                        name in ('o', 'f'))
            t[name] = e((8,) if unused else shape, dt)
        # the row-local chain kernel and its weight stream (repacked whenever the weights change: ensure_packed)
        self.chain = bool(chain) and self.want_fuse and bool(l.ttsmi_dense_chain_supported(cap, d, F))     # (training and forward-only plans)
        self.S = S
        self.chain_next, self.packed_ver, self.chain_on = None, None, False
        if self.chain:
            t['chain_w'] = e((int(l.ttsmi_dense_chain_pack_bytes(F, 1)),), torch.uint8)
        # ... and the backward's (csrc/chain16b.h: FFN dgrads + res-norm 1 backward + dctx as one launch; TTSMI_DENSE_CHAIN_BWD=0:
        # the three launches)
        self.chain_bwd = (self.chain and self.backward and _CHAIN_BWD and bool(l.ttsmi_dense_chain_bwd_supported(cap, d, F)))
        if self.chain_bwd:
            t['chain_bw'] = e((int(l.ttsmi_dense_chain_bwd_pack_bytes(F)),), torch.uint8)
        # the FFN's ReLU as one bit per element for the backward (ttsmi_dense_block.relu_bits); TTSMI_RELU_BITS=0: re-read h1
        self.relu_bits = backward and self.want_fuse and os.environ.get('TTSMI_RELU_BITS', '1') != '0'
        t['relu_bits'] = e((max(int(l.ttsmi_relu_bits_bytes(cap, F)), 8) if self.relu_bits else 8,), torch.uint8)
        # workspaces at their largest over every row count <= cap (tile heights switch with the row count)
        parts_cap = max((cap + 63) // 64 + 8, int(l.ttsmi_layernorm_bwd_xhat_nparts(cap)), int(l.ttsmi_hgemm_ln_bwd_nparts(cap)),
                        int(l.ttsmi_add_layernorm_bwd_nparts(cap)))
        ln_ws = int(max(l.ttsmi_add_layernorm_bwd_ws_bytes(cap, d), l.ttsmi_layernorm_partials_bytes(parts_cap, d)))
        t['ln_ws1'], t['ln_ws2'] = _ws(ln_ws, device), _ws(ln_ws, device)
        self.ln_ws_bytes = ln_ws
        if self.want_fuse:
            for name in ('xhat1', 'xhat2'):
                t[name] = e((cap, d), bf)
            t['lnp_ws1'] = _ws(l.ttsmi_layernorm_partials_bytes(parts_cap, d), device)
            # (res-norm 2's partials come from ttsmi_layernorm_bwd_xhat, or - chained - from the GEMM epilogue of the block above)
            t['lnp_ws2'] = _ws(l.ttsmi_layernorm_partials_bytes(parts_cap, d), device)
        key = (H, d, self.backward)
        if key not in shared or shared[key]['cap'] < cap:
            # (the model drops a stack's plans together when one of them has to grow, so nobody holds the old entry)
            shared[key] = ({'cap': cap, 'da': e((cap, d), f32), 'dctx': e((cap, d), bf),
                            'attn_ws': _ws(4 * cap * H + 1024, device)} if backward else
                           {'cap': cap, 'da': e((8,), f32), 'dctx': e((8,), bf),  # forward only: scratch of the split-key attention
                            'attn_ws': None})
        sh = shared[key]
        self.shared = sh
        self.wgrad_need = max(int(l.ttsmi_hgemm_wgrad_rows_ws_bytes(cap, kin, n))
                              for kin, n in ((F, d), (d, F), (d, d), (d, 3 * d)))
        self.events = [torch.cuda.Event() for _ in range(4)]
        for ev in self.events:
            ev.record()                                   # materialises the hipEvent_t behind the torch object
        D = self.desc = _lib.DenseBlockDesc()
        D.H, D.d, D.F = H, d, F
        for k, v in (('bqkv', P['bqkv']), ('bo', P['bo']), ('ln1_g', P['ln1.gamma']), ('ln1_b', P['ln1.beta']),
                     ('b1', P['ffn.b1']), ('b2', P['ffn.b2']), ('ln2_g', P['ln2.gamma']), ('ln2_b', P['ln2.beta']),
                     ('wqkv_t', S['wqkv'].wt), ('wo_t', S['wo'].wt), ('w1_t', S['ffn.w1'].wt), ('w2_t', S['ffn.w2'].wt),
                     ('wqkv_b', S['wqkv'].wb), ('wo_b', S['wo'].wb), ('w1_b', S['ffn.w1'].wb), ('w2_b', S['ffn.w2'].wb),
                     ('g_wqkv', G['wqkv']), ('g_bqkv', G['bqkv']), ('g_wo', G['wo']), ('g_bo', G['bo']),
                     ('g_ln1_g', G['ln1.gamma']), ('g_ln1_b', G['ln1.beta']), ('g_w1', G['ffn.w1']), ('g_b1', G['ffn.b1']),
                     ('g_w2', G['ffn.w2']), ('g_b2', G['ffn.b2']), ('g_ln2_g', G['ln2.gamma']), ('g_ln2_b', G['ln2.beta'])):
            assert v.is_contiguous()
            setattr(D, k, v.data_ptr())
        for k in ('qkv', 'cx', 'a_bf', 'h1', 'out_bf', 'lse', 'o', 'a', 'f', 'out', 'mean1', 'rstd1', 'mean2', 'rstd2',
                  'df', 'dh1', 'd_o', 'dqkv', 'dh', 'ln_ws1', 'ln_ws2'):
            setattr(D, k, t[k].data_ptr())
        if self.want_fuse:
            D.lnp_ws1_bytes, D.lnp_ws2_bytes = t['lnp_ws1'].numel(), t['lnp_ws2'].numel()
            for k in ('xhat1', 'xhat2', 'lnp_ws1', 'lnp_ws2'):
                setattr(D, k, t[k].data_ptr())
        D.da, D.dctx = sh['da'].data_ptr(), sh['dctx'].data_ptr()
        D.relu_bits = t['relu_bits'].data_ptr() if self.relu_bits else None
        D.ln_ws_bytes = ln_ws
        for i, ev in enumerate(self.events):
            D.ev[i] = ev.cuda_event
        self.G = G
        self.above = None
        self._dref = ctypes.byref(D)
        self.B = self.T = self.M = 0
        self.bound_by = None
        self.rebind(B, T)

    def rebind(self, B, T):
        """Point the plan at a batch of B x T rows (<= the capacity): shape fields, the row-count dependent workgroup
        counts and workspace sizes.  No allocation except the forward-only split-key scratch when it has to grow."""
        if (B, T) == (self.B, self.T):
            return
        l = _lib.lib()
        M = B * T
        assert M <= self.cap, (B, T, self.cap)
        H, d = self.H, self.d
        D, sh = self.desc, self.shared
        self.B, self.T, self.M = B, T, M
        D.B, D.T = B, T
        # LayerNorms fused into the GEMM epilogues (ttsmi_hgemm_ln_fwd / _bwd): needs the full row in one tile, so
        # few, tall workgroups - at inference with few rows (batch 1: 2304 rows = 18..36 workgroups, 29 us against
        # 9.5 + 5 us for GEMM + LayerNorm measured) the unfused pair is faster; training always fuses (the x-hat
        # backward is where most of the gain is)
        self.fuse_ln = self.want_fuse and (self.backward or M >= FUSE_LN_MIN_ROWS_INFERENCE)
        D.fuse_ln = int(self.fuse_ln)
        self.lnp_nw1_rowgemm, self.lnp_nw2 = int(l.ttsmi_hgemm_ln_bwd_nparts(M)), int(l.ttsmi_layernorm_bwd_xhat_nparts(M))
        self.lnp_nw1 = self.lnp_nw1_rowgemm           # (bind: the backward chain's count when that is what will run)
        if self.backward:
            need = int(l.ttsmi_attention_bwd_ws_bytes(B, H, T, d // H))
            assert need <= sh['attn_ws'].numel(), (need, sh['attn_ws'].numel())
        else:
            need = max(256, int(l.ttsmi_attention_fwd_splitkeys_ws_bytes(B, H, T, d // H)))
            if sh['attn_ws'] is None or sh['attn_ws'].numel() < need:
                # Every scratch ever bound stays alive on the stack-wide entry: inference graphs captured earlier hold the
                # old tensor's RAW pointer (their plans are shared objects whose `_attn_ws` moves on with this rebind), so
                # freeing it would let a later replay write attention partials into memory the allocator has handed to
                # somebody else (advisor finding, round 3).  The need is monotone per stack, so the list stays short.
                if sh['attn_ws'] is not None:
                    sh.setdefault('attn_ws_retired', []).append(sh['attn_ws'])
                sh['attn_ws'] = _ws(need, self.t['qkv'].device)
            self._attn_ws = sh['attn_ws']
        D.attn_ws, D.attn_ws_bytes = sh['attn_ws'].data_ptr(), sh['attn_ws'].numel()
        D.attn_split = int(not self.backward and os.environ.get('TTSMI_ATTN_SPLIT', '1') != '0' and
                           l.ttsmi_attention_fwd_splitkeys_ws_bytes(B, H, T, d // H) > 0)
        if self.above is not None and (self.above.B, self.above.T) != (B, T):
            self.chain_above(None)
        if self.chain_next is not None and (self.chain_next.B, self.chain_next.T) != (B, T):
            self.chain_forward(None)

    def chain_above(self, above):
        """`above` consumes this block's output and nothing else does: its backward finishes this block's res-norm-2
        backward in the epilogue of its last dgrad (ttsmi_dense_block.below).  None unlinks."""
        ok = (above is not None and self.fuse_ln and above.fuse_ln and self.backward and above.backward and
              (above.B, above.T, above.d) == (self.B, self.T, self.d))
        if self.above is not None and self.above is not above:
            self.above.desc.below = None
        self.above = above if ok else None
        self.desc.ln2_done = int(ok)
        if ok:
            above.desc.below = ctypes.addressof(self.desc)
        return ok

    def chain_forward(self, nxt):
        """FORWARD chaining (ttsmi_dense_block.above / qkv_done): `nxt` is the next block of the stack - this block's chain
        launch also computes nxt's qkv projection, and nxt skips its own.  None unlinks.  Takes effect only while this
        block's chain is on (bind decides per step)."""
        ok = (nxt is not None and self.chain and (nxt.B, nxt.T, nxt.d) == (self.B, self.T, self.d))
        if self.chain_next is not None and self.chain_next is not nxt:
            self.chain_next.desc.qkv_done = 0
        if (nxt if ok else None) is not self.chain_next:
            self.packed_ver = None                    # the stream's qkv tail belongs to another block (or goes away)
        self.chain_next = nxt if ok else None
        return ok

    def ensure_packed(self, version):
        """The chain's weight stream is current for weight version `version` (the model bumps it whenever the bf16
        shadows are refreshed).  Under stream capture the pack launch is always issued, so that a replayed step repacks."""
        if not self.chain_on:
            return
        ev, self.pack_ev = self.pack_ev, None
        if ev is not None:                        # packed ahead on a side stream (ForwardTransformer._launch_chain_packs): the
            cur_stream().wait_event(ev)           # consumer waits for it - and so does an in-line repack of the same buffers
        if self.packed_ver == version and not torch.cuda.is_current_stream_capturing():
            return
        S, nxt = self.S, self.chain_next
        check(_lib.lib().ttsmi_dense_chain_pack(_p(S['wo'].wt), _p(S['ffn.w1'].wt), _p(S['ffn.w2'].wt),
                                                _p(nxt.S['wqkv'].wt) if nxt is not None else None, self.F,
                                                _p(self.t['chain_w']), self.t['chain_w'].numel(), _stream()), 'dense_chain_pack')
        if self.chain_bwd:
            check(_lib.lib().ttsmi_dense_chain_bwd_pack(_p(S['ffn.w1'].wb), _p(S['ffn.w2'].wb), _p(S['wo'].wb), self.F,
                                                        _p(self.t['chain_bw']), self.t['chain_bw'].numel(), _stream()),
                  'dense_chain_bwd_pack')
        self.packed_ver = version

    def bind(self, pad, klen, rate, drop, sites, dmask, res16=False, out32=True):
        """Per-step inputs of the descriptor (masks are new tensors every step; the rest rarely changes).
        res16: bf16 residual stream between the fused kernels (ttsmi_dense_block.res16; ignored without fused LayerNorms);
        out32: the fp32 block output is read by somebody (the last block of a stack, activation taps)."""
        D = self.desc
        self.bound_by = None                              # (a TrainStepPlan that binds this plan stamps itself here afterwards)
        self.res16 = bool(res16) and self.fuse_ln
        D.res16 = (1 | (2 if out32 else 0)) if self.res16 else 0
        D.pad, D.klen = pad.data_ptr(), klen.data_ptr()
        D.rate, D.seed, D.step_dev = float(rate), drop.seed, _p(drop.step_dev)
        D.site_attn, D.site_ln1, D.site_ln2 = sites
        D.dropmask = _p(dmask)
        D.main_stream = _stream()
        self.keep = (pad, klen, dmask)                    # alive until the next bind
        # the row-local chain needs the bf16 residual stream, and pays from decoder-size batches on (measured, 16-row form:
        # 85.7 us against 104.7 for the four launches at 28 800 rows, level at 12 000, 66 against 57 at 6 400 - a workgroup
        # takes ~60 us whatever the row count); the link to the next block follows it
        self.chain_on = self.chain and self.res16 and self.M >= CHAIN_MIN_ROWS
        nxt = self.chain_next if self.chain_on else None
        D.chain_w = self.t['chain_w'].data_ptr() if self.chain_on else None
        D.chain_w_bytes = self.t['chain_w'].numel() if self.chain_on else 0
        D.above = ctypes.addressof(nxt.desc) if nxt is not None else None
        if self.chain_next is not None:
            self.chain_next.desc.qkv_done = int(nxt is not None)
        bw_on = self.chain_on and self.chain_bwd and self.relu_bits and torch.is_grad_enabled()
        D.chain_bw = self.t['chain_bw'].data_ptr() if bw_on else None
        D.chain_bw_bytes = self.t['chain_bw'].numel() if bw_on else 0
        # res-norm 1's parameter partials: one row per workgroup of whichever kernel will run its backward (the library's own
        # predicate - the backward chain has 128-row workgroups, the full-row GEMM 64 or 128 by row count)
        l = _lib.lib()
        self.lnp_nw1 = int(l.ttsmi_dense_chain_bwd_nparts(self.M) if (bw_on and l.ttsmi_dense_block_bwd_chained(self._dref))
                           else self.lnp_nw1_rowgemm)

    def fwd(self, h, h_bf):
        check(_lib.lib().ttsmi_dense_block_fwd(self._dref, _p(h), _p(h_bf)), 'dense_block_fwd')

    def _prepare_bwd(self, device, need=None):
        """The weight-gradient stream / workspace fields of the descriptor for the coming backward.  need: the largest
        `wgrad_need` of the plans that one C call will launch (a stack): the shared workspace is grown ONCE, before any
        descriptor of the stack takes its pointer - growing it between two plans would leave the earlier descriptor
        pointing at a tensor just handed back to the allocator."""
        assert self.backward, 'forward-only plan'
        D = self.desc
        need = self.wgrad_need if need is None else max(int(need), self.wgrad_need)
        if _WgradStream.enabled:
            W = _WgradStream.cur()
            if W.stream is None:
                W.stream = _new_wgrad_stream()
            if W.handle is None:
                W.handle = W.stream.cuda_stream
            if W.ws is None or W.ws.numel() < need:
                with torch.cuda.stream(W.stream):
                    W.ws = torch.empty(int(max(need, 1 << 26)), dtype=torch.uint8, device=device)
            D.side_stream, D.wgrad_ws, D.wgrad_ws_bytes = W.handle, W.ws.data_ptr(), W.ws.numel()
            W.pending = True
        else:
            ws = _ws(self.wgrad_need, device)
            self.keep = self.keep + (ws,)
            D.side_stream, D.wgrad_ws, D.wgrad_ws_bytes = None, ws.data_ptr(), ws.numel()

    def bwd(self, h, h_bf, dout):
        self._prepare_bwd(h.device if h is not None else h_bf.device)
        check(_lib.lib().ttsmi_dense_block_bwd(self._dref, _p(h), _p(h_bf), _p(dout)), 'dense_block_bwd')
        self._defer_ln()

    def _defer_ln(self):
        """The block's LayerNorm parameter-gradient partials join the step's batched reduction."""
        D = self.desc
        t, G, M, d = self.t, self.G, self.M, self.d

        def defer():
            if self.fuse_ln:       # partial rows left by ttsmi_layernorm_bwd_xhat / the epilogue of ttsmi_hgemm_ln_bwd
                _ln_defer(t['lnp_ws2'], G['ln2.gamma'], G['ln2.beta'], None, M, d,
                          self.lnp_nw1_rowgemm if D.ln2_done else self.lnp_nw2)
                _ln_defer(t['lnp_ws1'], G['ln1.gamma'], G['ln1.beta'], None, M, d, self.lnp_nw1)
            else:
                _ln_defer(t['ln_ws2'], G['ln2.gamma'], G['ln2.beta'], None, M, d)
                _ln_defer(t['ln_ws1'], G['ln1.gamma'], G['ln1.beta'], None, M, d)
        if _LN_PENDING is not None:
            defer()
        else:
            with ln_param_batch():
                defer()


class PlannedDenseBlockFn(torch.autograd.Function):
    """DenseBlockFn on a DenseBlockPlan: forward / backward are one C++ call each; the block's activations live in
    the plan (valid until the plan's next forward - ForwardTransformer uses it inside one train step only)."""

    @staticmethod
    def forward(ctx, h, h_bf, plan):
        plan.fwd(h, h_bf)
        ctx.plan, ctx.h, ctx.h_bf = plan, h, h_bf
        M = plan.M                                                              # fresh aliases of the persistent buffers' live rows
        out, out_bf = plan.t['out'][:M].detach(), plan.t['out_bf'][:M].detach()
        ctx.mark_non_differentiable(out_bf)
        ctx.set_materialize_grads(False)
        return out, out_bf

    @staticmethod
    def backward(ctx, dout, _dout_bf):
        plan = ctx.plan
        plan.bwd(ctx.h, ctx.h_bf, _c(dout))
        return plan.t['dh'][:plan.M].detach(), None, None


class PlannedDenseStackFn(torch.autograd.Function):
    """A whole stack of consecutive planned dense blocks as ONE autograd node: ttsmi_dense_stack_fwd / _bwd issue every
    block's launches from one C++ call each.  Same launches in the same order as a PlannedDenseBlockFn per block (results
    are bit-identical); what disappears is ~10 autograd nodes and ~40 Python -> C round trips per step - with the
    reference's bucketed batches (~12 k rows) the step is bound by the host's issue rate (bench.py --workload lj-dist)."""

    @staticmethod
    def forward(ctx, h, h_bf, plans):
        n = len(plans)
        arr = (ctypes.c_void_p * n)(*[ctypes.addressof(pl.desc) for pl in plans])
        check(_lib.lib().ttsmi_dense_stack_fwd(arr, n, _p(h), _p(h_bf)), 'dense_stack_fwd')
        ctx.plans, ctx.arr, ctx.h, ctx.h_bf = plans, arr, h, h_bf
        top = plans[-1]
        out, out_bf = top.t['out'][:top.M].detach(), top.t['out_bf'][:top.M].detach()
        ctx.mark_non_differentiable(out_bf)
        ctx.set_materialize_grads(False)
        return out, out_bf

    @staticmethod
    def backward(ctx, dout, _dout_bf):
        plans = ctx.plans
        dev = ctx.h_bf.device
        need = max(pl.wgrad_need for pl in plans)
        for pl in plans:
            pl._prepare_bwd(dev, need)
        check(_lib.lib().ttsmi_dense_stack_bwd(ctx.arr, len(plans), _p(ctx.h), _p(ctx.h_bf), _p(_c(dout))), 'dense_stack_bwd')
        for pl in reversed(plans):
            pl._defer_ln()
        bottom = plans[0]
        return bottom.t['dh'][:bottom.M].detach(), None, None
