"""Measurement build only (tools/build_variant.sh abl "-DTTSMI_ABLATION_BUILD" chain.hip ...): where a workgroup of the chain
kernel spends its cycles - per phase stamps of every (workgroup, wave), and the launch time under the stage ablations."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import _lib, ops          # noqa: E402
from transformertts_amd.ops import _p, _stream, check     # noqa: E402

DEV, D, F, EPS = 'cuda:0', 256, 1024, 1e-6
l = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 28800
g = lambda *s, sc=1.0: torch.randn(*s, device=DEV) * sc
h, cx = g(M, D).bfloat16(), g(M, D).bfloat16()
sh = {k: ops.make_shadow(v) for k, v in dict(wo=g(2 * D, D, sc=0.05), w1=g(D, F, sc=0.06), w2=g(F, D, sc=0.04), wq=g(D, 3 * D, sc=0.06)).items()}
bo, b1, b2, bq, g1, be1, g2, be2 = g(D), g(F), g(D), g(3 * D), 1 + 0.1 * g(D), 0.1 * g(D), 1 + 0.1 * g(D), 0.1 * g(D)
pad = (torch.arange(M, device=DEV) % 11 == 4).to(torch.uint8)
step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
nb = int(l.ttsmi_dense_chain_pack_bytes(F, 1))
wpack = torch.empty(nb, dtype=torch.uint8, device=DEV)
check(l.ttsmi_dense_chain_pack(_p(sh['wo'].wt), _p(sh['w1'].wt), _p(sh['w2'].wt), _p(sh['wq'].wt), F, _p(wpack), nb, _stream()))
e = lambda *s, dt=torch.bfloat16: torch.empty(s, dtype=dt, device=DEV)
a, xh1, r1, h1, o, xh2, r2, qkv = e(M, D), e(M, D), e(M, dt=torch.float32), e(M, F), e(M, D), e(M, D), e(M, dt=torch.float32), e(M, 3 * D)
NWV = 4 if (M <= 16384 and os.environ.get('TTSMI_DENSE_CHAIN_NW', '0') != '8') or os.environ.get('TTSMI_DENSE_CHAIN_NW') == '4' else 8      # waves per workgroup (csrc/chain.hip: chain_nw)
nwg = (M + 16 * NWV - 1) // (16 * NWV)
if NWV == 4 and M <= 8192 and os.environ.get('TTSMI_DENSE_CHAIN_SPLIT', '1') != '0':
    nwg = (nwg + 7) // 8 * 16                      # the SPLIT form: two workgroups per tile, groups of 16
dbg = torch.zeros(nwg * NWV * 8, dtype=torch.int64, device=DEV)
if hasattr(l._cdll, 'ttsmi_dense_chain_debug'):
    l._cdll.ttsmi_dense_chain_debug(ctypes.c_void_p(dbg.data_ptr()))


def chain():
    check(l.ttsmi_dense_chain_fwd(_p(h), _p(cx), _p(wpack), nb, M, F, _p(bo), _p(g1), _p(be1), _p(b1), _p(b2), _p(g2), _p(be2), _p(bq),
                                  _p(pad), 0.1, 99, _p(step), 5, 6, EPS, _p(a), _p(xh1), _p(r1), _p(h1), None, 0,
                                  _p(o), _p(xh2), _p(r2), None, _p(qkv), _stream()))


for _ in range(5):
    chain()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    chain()
e1.record()
torch.cuda.synchronize()
d = dbg.cpu().reshape(nwg, NWV, 8).double()
names = ['prologue (X / parameter loads, first DMAs)', 'o-projection (8 stages)', 'LayerNorm 1 + stores', 'FFN (32 stages)', 'LayerNorm 2 + stores',
         'qkv (12 stages)']
print(f'M={M} TTSMI_CHAIN_ABLATE={os.environ.get("TTSMI_CHAIN_ABLATE", "0")}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per launch')
tot = d[:, :, 1:7].sum(-1)
print(f'  cycles per wave: mean {tot.mean():.0f}  max {tot.max():.0f}   (in stage waits: mean {d[:, :, 7].mean():.0f})')
for i, n in enumerate(names):
    print(f'  {n:45s} mean {d[:, :, 1 + i].mean():8.0f}  max {d[:, :, 1 + i].max():8.0f}')
start = d[:, :, 0]
print(f'  workgroup start spread: {start.max() - start.min():.0f} cycles; per-wave means of total: {[round(float(tot[:, w].mean())) for w in range(NWV)]}')
