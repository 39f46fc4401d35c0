"""Two streams launching the SPLIT chain forms at the same time (each with its own exchange buffers, chain.hip:
chain_split_buffers): 2 x 208 workgroups compete for 256 CUs, so pairs of one launch wait for CUs the other holds.  Checks that
nothing hangs (a partner that never arrives traps after seconds, chain16.h: c16_wait_flag) and that every launch's outputs are
bit-identical to the same launch run alone.  Run under `timeout`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformertts_amd import _lib, ops          # noqa: E402
from transformertts_amd.ops import _p, check     # noqa: E402

DEV, D, F, EPS = 'cuda:0', 256, 1024, 1e-6
l = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6400
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 50


def make(seed):
    gen = torch.Generator(device='cpu').manual_seed(seed)
    g = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(DEV)
    t = dict(h=g(M, D).bfloat16(), cx=g(M, D).bfloat16())
    t['sh'] = {k: ops.make_shadow(v) for k, v in dict(wo=g(2 * D, D, sc=0.05), w1=g(D, F, sc=0.06), w2=g(F, D, sc=0.04), wq=g(D, 3 * D, sc=0.06)).items()}
    for k, n in dict(bo=D, b1=F, b2=D, bq=3 * D, be1=D, be2=D).items():
        t[k] = 0.1 * g(n)
    t['g1'], t['g2'] = 1 + 0.1 * g(D), 1 + 0.1 * g(D)
    t['pad'] = (torch.arange(M, device=DEV) % 11 == 4).to(torch.uint8)
    t['step'] = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    nb = int(l.ttsmi_dense_chain_pack_bytes(F, 1))
    t['nb'], t['wpack'] = nb, torch.empty(nb, dtype=torch.uint8, device=DEV)
    check(l.ttsmi_dense_chain_pack(_p(t['sh']['wo'].wt), _p(t['sh']['w1'].wt), _p(t['sh']['w2'].wt), _p(t['sh']['wq'].wt), F, _p(t['wpack']), nb,
                                   torch.cuda.current_stream().cuda_stream))
    return t


def outs():
    e = lambda *s, dt=torch.bfloat16: torch.zeros(s, dtype=dt, device=DEV)
    return dict(a=e(M, D), xh1=e(M, D), r1=e(M, dt=torch.float32), h1=e(M, F), o=e(M, D), xh2=e(M, D), r2=e(M, dt=torch.float32), qkv=e(M, 3 * D))


def chain(t, o, stream):
    check(l.ttsmi_dense_chain_fwd(_p(t['h']), _p(t['cx']), _p(t['wpack']), t['nb'], M, F, _p(t['bo']), _p(t['g1']), _p(t['be1']), _p(t['b1']), _p(t['b2']),
                                  _p(t['g2']), _p(t['be2']), _p(t['bq']), _p(t['pad']), 0.1, 99, _p(t['step']), 5, 6, EPS, _p(o['a']), _p(o['xh1']),
                                  _p(o['r1']), _p(o['h1']), None, 0, _p(o['o']), _p(o['xh2']), _p(o['r2']), None, _p(o['qkv']), stream.cuda_stream))


A, B = make(1), make(2)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ref = []
for t, s in ((A, s1), (B, s2)):                  # each alone (this also allocates the two streams' exchange buffers)
    o = outs()
    chain(t, o, s)
    torch.cuda.synchronize()
    ref.append({k: v.clone() for k, v in o.items()})
print('route:', l.ttsmi_last_kernel().decode())
oa, ob = outs(), outs()
bad = 0
for r in range(ROUNDS):
    for v in list(oa.values()) + list(ob.values()):
        v.zero_()
    torch.cuda.synchronize()
    for _ in range(4):                            # eight launches in flight, interleaved on the two queues
        chain(A, oa, s1)
        chain(B, ob, s2)
    torch.cuda.synchronize()
    for o, want in ((oa, ref[0]), (ob, ref[1])):
        for k in want:
            if not torch.equal(o[k], want[k]):
                bad += 1
                print(f'round {r}: {k} differs, max abs {float((o[k].float() - want[k].float()).abs().max()):.3e}')
print(f'M={M}: {ROUNDS} rounds of 2 x 4 concurrent launches: {"all outputs bit-identical to the launches run alone" if bad == 0 else f"{bad} MISMATCHES"}')
sys.exit(1 if bad else 0)
