#!/usr/bin/env python
"""Where does a WAVE of the attention forward spend its time?  Measurement build only:
    tools/build_variant.sh abl "-DTTSMI_ABLATION_BUILD" dense_block.hip attention_bf16.hip
    TTSMI_ALLOW_LIB_OVERRIDE=1 TTSMI_LIB=transformertts_amd/lib/libttsmi_abl.so python tools/debug/attn_fwd_sections.py
Every wave stamps s_memtime at the section boundaries of its loop (the stamps first drain the wave's LDS / scalar
counters, so they serialise a little: the launch is ~10 % slower than unstamped); printed: shader cycles per 32 x 32 score
block, averaged over the waves that had work, for the benchmark's decoder shape."""
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
B, H, T, dh, pdrop = 32, 4, int(os.environ.get('T', '900')), 64, float(os.environ.get('PDROP', '0.1'))
d = H * dh
qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.5).to(torch.bfloat16)
pad = torch.zeros(B, T, dtype=torch.uint8, device=dev)
klen = torch.full((B,), T, dtype=torch.int32, device=dev)
ctx = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, T, device=dev)
step = torch.zeros(1, dtype=torch.int64, device=dev)
drop = ops.DropCtx(7, step)
dm = ops.attention_dropmask(B, H, T, pdrop, drop, 3, dev) if pdrop > 0 else None


def launch():
    if pdrop > 0:
        check(l.ttsmi_attention_fwd_masked(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, pdrop, _p(dm), _lib.TTSMI_BF16_IO, _stream()))
    else:
        check(l.ttsmi_attention_fwd(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, 0.0, 7, _p(step), 3, _lib.TTSMI_BF16_IO, _stream()))


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    launch()
e1.record()
torch.cuda.synchronize()
fn = l._cdll.ttsmi_debug_fused_dump
fn.restype = ctypes.c_int
buf = np.zeros((8192, 8), dtype=np.uint64)
n = fn(buf.ctypes.data_as(ctypes.c_void_p), 8192)
assert n > 0, 'not a measurement build'
a = buf[:n].astype(np.float64)
live = a[a[:, 5] > 0]
blocks = live[:, 5]
names = ['barriers + stash (per 64-key tile: / 2)', 'fetch issue', 'S = Q.K^T up to first use', 'softmax arithmetic', 'P.V product']
print(f'{e0.elapsed_time(e1) * 100:.1f} us per launch (stamped); {len(live)} waves with work of {n}; blocks per wave {blocks.mean():.1f}; '
      f'wave lifetime {live[:, 6].mean():.0f} cycles = {live[:, 6].mean() / blocks.mean():.0f} per block')
for i, nm in enumerate(names):
    print(f'  {nm:42s} {(live[:, i] / blocks).mean():8.0f} cycles per block')
print(f'  {"sum of the sections":42s} {(live[:, :5].sum(1) / blocks).mean():8.0f}')
e = (live[:, 7] / blocks).mean()
label = 'two stamps back to back (one stamp alone)'
net = ', '.join(f'{(live[:, i] / blocks).mean() - e * (0.5 if i < 2 else 1.0):.0f}' for i in range(5))
print(f'  {label:42s} {e:8.0f}   -> sections net of one stamp each: {net}')
