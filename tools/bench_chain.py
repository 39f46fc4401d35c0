"""Time the row-local chain kernel alone (ttsmi_dense_chain_fwd) against the four launches it replaces
(ttsmi_hgemm_ln_fwd_h, ttsmi_hgemm_k256_relu_bits, ttsmi_hgemm_ln_fwd_h, ttsmi_hgemm_tn) on the same tensors."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import _lib, ops          # noqa: E402
from transformertts_amd.ops import _p, _stream, check     # noqa: E402

DEV, D, F, EPS = 'cuda:0', 256, 1024, 1e-6
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


for M in [int(a) for a in sys.argv[1:]] or [28800, 6400, 12000, 2500]:
    g = lambda *s, sc=1.0: torch.randn(*s, device=DEV) * sc
    h, cx = g(M, D).bfloat16(), g(M, D).bfloat16()
    sh = {k: ops.make_shadow(v) for k, v in dict(wo=g(2 * D, D, sc=0.05), w1=g(D, F, sc=0.06), w2=g(F, D, sc=0.04), wq=g(D, 3 * D, sc=0.06)).items()}
    bo, b1, b2, bq, g1, be1, g2, be2 = g(D), g(F), g(D), g(3 * D), 1 + 0.1 * g(D), 0.1 * g(D), 1 + 0.1 * g(D), 0.1 * g(D)
    pad = (torch.arange(M, device=DEV) % 11 == 4).to(torch.uint8)
    step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    nb = int(l.ttsmi_dense_chain_pack_bytes(F, 1))
    wpack = torch.empty(nb, dtype=torch.uint8, device=DEV)
    pack = lambda: check(l.ttsmi_dense_chain_pack(_p(sh['wo'].wt), _p(sh['w1'].wt), _p(sh['w2'].wt), _p(sh['wq'].wt), F, _p(wpack), nb, _stream()))
    pack()
    e = lambda *s, dt=torch.bfloat16: torch.empty(s, dtype=dt, device=DEV)
    a, xh1, r1, h1, o, xh2, r2, qkv = e(M, D), e(M, D), e(M, dt=torch.float32), e(M, F), e(M, D), e(M, D), e(M, dt=torch.float32), e(M, 3 * D)
    bits = torch.zeros(int(l.ttsmi_relu_bits_bytes(M, F)), dtype=torch.uint8, device=DEV)
    use_bits = bool(l._cdll.ttsmi_hgemm_k256_eligible(M, F, D))

    def chain():
        check(l.ttsmi_dense_chain_fwd(_p(h), _p(cx), _p(wpack), nb, M, F, _p(bo), _p(g1), _p(be1), _p(b1), _p(b2), _p(g2), _p(be2), _p(bq),
                                      _p(pad), 0.1, 99, _p(step), 5, 6, EPS, _p(a), _p(xh1), _p(r1), _p(h1), _p(bits) if use_bits else None, 0,
                                      _p(o), _p(xh2), _p(r2), None, _p(qkv), _stream()))

    def four():
        check(l.ttsmi_hgemm_ln_fwd_h(_p(h), D, _p(cx), D, D, _p(sh['wo'].wt), 2 * D, _p(bo), _p(h), _p(g1), _p(be1), _p(pad), 0.1, 5, 99, _p(step),
                                     EPS, None, _p(a), _p(xh1), _p(r1), M, D, 2 * D, _stream()))
        if use_bits:
            check(l.ttsmi_hgemm_k256_relu_bits(_p(a), D, _p(sh['w1'].wt), D, _p(b1), _p(h1), F, _p(bits), M, F, _stream()))
        else:
            check(l.ttsmi_hgemm_tn(_p(a), 0, D, None, 0, 0, _p(sh['w1'].wt), D, _p(b1), None, 0, _p(h1), F, M, F, D, 1 | 4, 1, 0, 0, 0, _stream()))
        check(l.ttsmi_hgemm_ln_fwd_h(_p(h1), F, None, 0, 0, _p(sh['w2'].wt), F, _p(b2), _p(a), _p(g2), _p(be2), _p(pad), 0.1, 6, 99, _p(step),
                                     EPS, None, _p(o), _p(xh2), _p(r2), M, D, F, _stream()))
        check(l.ttsmi_hgemm_tn(_p(o), 0, D, None, 0, 0, _p(sh['wq'].wt), D, _p(bq), None, 0, _p(qkv), 3 * D, M, 3 * D, D, 4, 1, 0, 0, 0, _stream()))

    t_pack, t_chain = timed(pack), timed(chain)
    try:
        t_four = timed(four)
    except Exception as ex:            # noqa: BLE001
        t_four = float('nan')
        print('four-launch form failed:', ex)
    fl = 2.0 * M * D * (2 * D + 2 * F + 3 * D)
    print(f'M={M:6d}  chain {t_chain:7.1f} us ({fl / t_chain / 1e6:6.1f} TF)   four launches {t_four:7.1f} us   pack {t_pack:5.1f} us')
