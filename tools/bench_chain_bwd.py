"""Time the backward chain kernel alone (ttsmi_dense_chain_bwd) against the three launches it replaces
(ttsmi_hgemm_k256_masked_bits, ttsmi_hgemm_ln_bwd_dual_h, the dctx GEMM) on the same tensors."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import _lib, ops          # noqa: E402
from transformertts_amd.ops import _p, _stream, check     # noqa: E402

DEV, D, F = 'cuda:0', 256, 1024
l = _lib.lib()


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for M in [int(a) for a in sys.argv[1:]] or [28800, 16384]:
    g = lambda *s, sc=1.0: torch.randn(*s, device=DEV) * sc
    df, da, xh = g(M, D, sc=0.5).bfloat16(), g(M, D, sc=0.5).bfloat16(), g(M, D).bfloat16()
    rstd, gam = 0.5 + g(M).abs(), 1 + 0.1 * g(D)
    sh = {k: ops.make_shadow(v) for k, v in dict(w1=g(D, F, sc=0.06), w2=g(F, D, sc=0.04), wo=g(2 * D, D, sc=0.05)).items()}
    pad = (torch.arange(M, device=DEV) % 11 == 4).to(torch.uint8)
    step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    nb = int(l.ttsmi_dense_chain_bwd_pack_bytes(F))
    wpack = torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(l.ttsmi_dense_chain_bwd_pack(_p(sh['w1'].wb), _p(sh['w2'].wb), _p(sh['wo'].wb), F, _p(wpack), nb, _stream()))
    bits_lane = torch.randint(0, 256, (int(l.ttsmi_relu_bits_bytes(M, F)) + 4096,), dtype=torch.uint8, device=DEV)
    bits_k256 = torch.randint(0, 256, (int(l.ttsmi_relu_bits_bytes(M, F)),), dtype=torch.uint8, device=DEV)
    e = lambda *s, dt=torch.bfloat16: torch.empty(s, dtype=dt, device=DEV)
    dh1, d_o, dres, dctx = e(M, F), e(M, D), e(M, D), e(M, D)
    nparts = int(l.ttsmi_dense_chain_bwd_nparts(M))
    part = ops._ws(int(l.ttsmi_layernorm_partials_bytes(max(nparts, int(l.ttsmi_hgemm_ln_bwd_nparts(M))), D)), DEV)

    def chain():
        check(l.ttsmi_dense_chain_bwd(_p(df), _p(da), _p(xh), _p(rstd), _p(gam), _p(pad), _p(bits_lane), _p(wpack), nb, M, F, 0.1, 99, _p(step), 5,
                                      _p(dh1), _p(d_o), _p(dres), 1, _p(dctx), _p(part), part.numel(), _stream()))

    def three():
        check(l.ttsmi_hgemm_k256_masked_bits(_p(df), D, _p(sh['w2'].wb), D, _p(bits_k256), _p(dh1), F, M, F, _stream()))
        check(l.ttsmi_hgemm_ln_bwd_dual_h(_p(dh1), F, None, 0, 0, _p(sh['w1'].wb), F, None, 0, _p(da), _p(xh), _p(rstd), _p(gam), _p(pad), 0.1, 5, 99,
                                          _p(step), _p(d_o), _p(dres), 1, _p(part), part.numel(), M, D, F, _stream()))
        check(l.ttsmi_hgemm_tn(_p(d_o), 0, D, None, 0, 0, _p(sh['wo'].wb[D:]), D, None, None, 0, _p(dctx), D, M, D, D, 4, 1, 0, 0, 0, _stream()))

    t_chain, t_three = timed(chain), timed(three)
    fl = 2.0 * M * D * (2 * F + D)
    print(f'M={M:6d}  backward chain {t_chain:7.1f} us ({fl / t_chain / 1e6:6.1f} TF)   three launches {t_three:7.1f} us')
