#!/usr/bin/env python
"""Where does a workgroup of the one-pass attention backward spend its life?  Needs a measurement build:
    tools/build_variant.sh abl "-DTTSMI_ABLATION_BUILD" attention_bf16.hip
    TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=transformertts_amd/lib/libttsmi_abl.so python tools/debug/fused_bwd_timeline.py
Launches the kernel at the benchmark shape and prints, per key tile (= position in the hand-off chain): start time after
the first workgroup, lifetime, time spent waiting for hand-offs, time from start to the first hand-off, polls."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformertts_amd import _lib, ops  # noqa: E402
from transformertts_amd.ops import _p, _stream, check  # noqa: E402

l = _lib.lib()
dev = 'cuda:0'
B, H, T, dh, pdrop = 32, 4, 900, 64, float(os.environ.get('PDROP', '0.1'))
d = H * dh
qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
dctx = (torch.randn(B * T, d, device=dev) * 0.1).to(torch.bfloat16)
pad = torch.zeros(B, T, dtype=torch.uint8, device=dev)
klen = torch.full((B,), T, dtype=torch.int32, device=dev)
ctx = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
dqkv = torch.empty_like(qkv)
step = torch.zeros(1, dtype=torch.int64, device=dev)
drop = ops.DropCtx(7, step)
dm = ops.attention_dropmask(B, H, T, pdrop, drop, 3, dev) if pdrop > 0 else None
if pdrop > 0:
    check(l.ttsmi_attention_fwd_masked(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, pdrop, _p(dm), _lib.TTSMI_BF16_IO, _stream()))
else:
    check(l.ttsmi_attention_fwd(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, 0.0, 7, _p(step), 3, _lib.TTSMI_BF16_IO, _stream()))
fws = torch.empty(int(l.ttsmi_attention_bwd_fused_ws_bytes(B, H, T)), dtype=torch.uint8, device=dev)
check(l.ttsmi_attention_bwd_fused_ws_init(_p(fws), fws.numel(), _stream()))
for _ in range(3):
    check(l.ttsmi_attention_bwd_fused(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(dctx), _p(lse), _p(dqkv), B, H, T, dh, pdrop, 0, None, 0,
                                      _p(dm), _p(fws), fws.numel(), _stream()))
torch.cuda.synchronize()
fn = l._cdll.ttsmi_debug_fused_dump
fn.restype = ctypes.c_int
buf = np.zeros((8192, 8), dtype=np.uint64)
n = fn(buf.ctypes.data_as(ctypes.c_void_p), 8192)
assert n > 0, 'not a measurement build (ttsmi_debug_fused_dump returned %d)' % n
a = buf[:n].astype(np.int64)
t0 = a[:, 0].min()
us = lambda x: x / 100.0                       # 100 MHz
print(f'{n} workgroups, kernel span {us(a[:, 1].max() - t0):.1f} us; diag {fws[:8].view(torch.int32).cpu().tolist()}')
print('key tile |  start after first (mean / max) | lifetime | waiting | to first hand-off | polls   [us]')
for j in range(int(a[:, 5].max()) + 1):
    m = a[a[:, 5] == j]
    print(f'   {j}     | {us((m[:, 0] - t0).mean()):8.1f} {us((m[:, 0] - t0).max()):8.1f}        | {us((m[:, 1] - m[:, 0]).mean()):7.1f}  | {us(m[:, 2].mean()):6.1f}  |'
          f' {us(m[:, 3].mean()):8.1f}          | {m[:, 4].mean():7.1f}')
# dispatch order inside an XCD: how often did block b start AFTER block b + 8?
inv = sum(1 for b in range(n - 8) if a[b, 0] > a[b + 8, 0])
print(f'start-order inversions inside an XCD: {inv} of {n - 8}')
first = a[:512]
print(f'first 512 block ids: start spread {us(first[:, 0].max() - t0):.1f} us; block ids >= 512: first start {us(a[512:, 0].min() - t0):.1f} us')
