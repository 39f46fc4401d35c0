#!/usr/bin/env python
"""Time ttsmi_attention_weights: fp32 recomputation vs the bf16 kernel, with and without dropout (decoder shape)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import ops, _lib
B, H, T, dh = 32, 4, 900, 64
d = H * dh
qkv = (torch.randn(B * T, 3 * d, device='cuda') * 0.5).bfloat16()
q32 = qkv.float()
pad = torch.zeros(B, T, dtype=torch.uint8, device='cuda')
lse = torch.randn(B, H, T, device='cuda') + 7
step = torch.zeros(1, dtype=torch.int64, device='cuda')
drop = ops.DropCtx(seed=5, step_dev=step)
w = torch.empty(B, H, T, T, device='cuda')
l = _lib.lib()
from transformertts_amd.ops import _p, _stream, check
def run(x, dt, p):
    check(l.ttsmi_attention_weights(_p(x), _p(pad), _p(lse), _p(w), B, H, T, dh, p, 5, _p(step), 3, dt, _stream()))
for name, x, dt in (('fp32', q32, _lib.TTSMI_F32), ('bf16', qkv, _lib.TTSMI_BF16_IO)):
    for p in (0.0, 0.1):
        for _ in range(3): run(x, dt, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(x, dt, p)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        print(f'{name} p={p}: {t:.1f} us  ({w.numel() * 4 / t / 1e6:.2f} TB/s written)')
