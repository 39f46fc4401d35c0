"""-m gpu parity tests of every libttsmi op, called through the C ABI, against fp64 torch-CPU
restatements of the same op (oracle/ft_oracle.py building blocks).  Tolerances: integer / index
work bit-exact; fp32 work 1e-4 relative to the tensor's max magnitude (north_star), usually far
tighter because the fp32 MFMA path is an exact fp32 FMA chain."""
import math

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _ops():
    from transformertts_amd import ops
    return ops


def rel_err(got: torch.Tensor, want: torch.Tensor) -> float:
    want = want.double().cpu()
    got = got.double().cpu()
    scale = max(float(want.abs().max()), 1e-30)
    return float((got - want).abs().max()) / scale


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=gen) * scale).float()


# ------------------------------------------------------------------------------------ GEMM family
@pytest.mark.parametrize('M,K,N', [(128, 256, 256), (200, 64, 192), (1, 16, 4), (333, 100, 60),
                                   (77, 226, 1), (1000, 512, 130), (129, 17, 33)])
@pytest.mark.parametrize('relu', [False, True])
def test_linear_fwd(M, K, N, relu):
    ops = _ops()
    x, w, b = g(M, K, seed=1), g(K, N, seed=2), g(N, seed=3)
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), relu)
    want = x.double() @ w.double() + b.double()
    if relu:
        want = want.relu()
    assert rel_err(y, want) < 2e-6


def test_linear_fwd_asymmetric_identity():
    """A = I with an asymmetric B catches a transposed C write (guide rule 16)."""
    ops = _ops()
    n = 160
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97) - 3.0 * torch.arange(n)[None, :]
    y = ops.linear_fwd(torch.eye(n).to(DEV), w.to(DEV), None, False)
    assert torch.equal(y.cpu(), w)


@pytest.mark.parametrize('M,K1,K2,N', [(300, 64, 64, 64), (129, 256, 256, 256), (50, 16, 48, 20)])
def test_linear_fwd_dual_a(M, K1, K2, N):
    ops = _ops()
    x, x2, w, b = g(M, K1, seed=1), g(M, K2, seed=4), g(K1 + K2, N, seed=2), g(N, seed=3)
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), False, x2.to(DEV))
    want = torch.cat([x, x2], 1).double() @ w.double() + b.double()
    assert rel_err(y, want) < 2e-6


def test_linear_fwd_strided_input():
    ops = _ops()
    big = g(100, 192, seed=5).to(DEV)
    x = big[:, 64:128]                      # row stride 192, offset keeps 16-byte alignment
    w = g(64, 32, seed=6)
    y = ops.linear_fwd(x, w.to(DEV), None, False)
    assert rel_err(y, big.cpu()[:, 64:128].double() @ w.double()) < 2e-6


@pytest.mark.parametrize('M,K,N', [(256, 128, 256), (1000, 64, 192), (333, 100, 60), (77, 226, 1),
                                   (4000, 256, 1024)])
def test_linear_backward(M, K, N):
    ops = _ops()
    x, w, dy, h = g(M, K, seed=1), g(K, N, seed=2), g(M, N, seed=3), g(M, K, seed=4)
    dx = ops.linear_dgrad(dy.to(DEV), w.to(DEV))
    assert rel_err(dx, dy.double() @ w.double().T) < 2e-6
    dxm = ops.linear_dgrad(dy.to(DEV), w.to(DEV), relu_src=h.to(DEV))
    assert rel_err(dxm, (dy.double() @ w.double().T) * (h > 0)) < 2e-6
    dw = torch.full((K, N), 7.0, device=DEV)
    db = torch.full((N,), 7.0, device=DEV)
    ops.linear_wgrad(x.to(DEV), dy.to(DEV), dw, db)
    assert rel_err(dw, x.double().T @ dy.double()) < 3e-6
    assert rel_err(db, dy.double().sum(0)) < 3e-6


@pytest.mark.parametrize('B,T,Cin,Cout,k', [(2, 50, 64, 64, 3), (3, 17, 32, 226, 3), (2, 40, 226, 8, 3),
                                            (1, 9, 4, 4, 5), (2, 30, 16, 24, 1), (4, 200, 256, 256, 3),
                                            (2, 13, 12, 20, 4)])
def test_conv1d_fwd_bwd(B, T, Cin, Cout, k):
    ops = _ops()
    x, w, b = g(B, T, Cin, seed=1), g(k, Cin, Cout, seed=2, scale=0.2), g(Cout, seed=3)
    xd, wd, bd = (t.double().requires_grad_() for t in (x, w, b))
    want = fo.conv1d_same(xd, wd, bd).relu()
    y = ops.conv1d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), relu=True)
    assert rel_err(y, want.detach()) < 2e-6
    # backward of the pre-activation conv with an upstream gradient dy
    dy = g(B, T, Cout, seed=4)
    pre = fo.conv1d_same(xd, wd, bd)
    pre.backward(dy.double())
    dx = ops.conv1d_dgrad(dy.to(DEV), w.to(DEV))
    assert rel_err(dx, xd.grad) < 3e-6
    dw = torch.empty_like(w, device=DEV)
    db = torch.empty(Cout, device=DEV)
    ops.conv1d_wgrad(x.to(DEV), dy.to(DEV), dw, db)
    assert rel_err(dw, wd.grad) < 3e-6
    assert rel_err(db, bd.grad) < 3e-6
    hmask = g(B, T, Cin, seed=5)
    dxm = ops.conv1d_dgrad(dy.to(DEV), w.to(DEV), relu_src=hmask.to(DEV))
    assert rel_err(dxm, xd.grad * (hmask > 0)) < 3e-6


# ---- the same family on three bf16 MFMAs per product (dtype TTSMI_BF16X3): ~2^-16 per product instead of fp32's 2^-24
X3_TOL = 4e-5


@pytest.mark.parametrize('M,K,N', [(128, 256, 256), (200, 64, 192), (1, 16, 4), (333, 100, 60), (77, 226, 1), (1000, 512, 130), (129, 17, 33),
                                   (4000, 256, 1024)])
def test_bf16x3_linear_family(M, K, N):
    ops = _ops()
    from transformertts_amd import _lib
    X3 = _lib.TTSMI_BF16X3
    x, w, b, dy, h = g(M, K, seed=1), g(K, N, seed=2), g(N, seed=3), g(M, N, seed=4), g(M, K, seed=5)
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), True, dtype=X3)
    assert _lib.lib().ttsmi_last_kernel().decode() == 'gemm_x3_kernel'
    assert rel_err(y, (x.double() @ w.double() + b.double()).relu()) < X3_TOL
    dx = ops.linear_dgrad(dy.to(DEV), w.to(DEV), relu_src=h.to(DEV), dtype=X3)
    assert rel_err(dx, (dy.double() @ w.double().T) * (h > 0)) < X3_TOL
    acc = torch.ones(M, K, device=DEV)
    ops.linear_dgrad(dy.to(DEV), w.to(DEV), out=acc, accumulate=True, dtype=X3)
    assert rel_err(acc, dy.double() @ w.double().T + 1.0) < X3_TOL
    dw, db = torch.full((K, N), 7.0, device=DEV), torch.full((N,), 7.0, device=DEV)
    ops.linear_wgrad(x.to(DEV), dy.to(DEV), dw, db, dtype=X3)
    assert rel_err(dw, x.double().T @ dy.double()) < X3_TOL
    assert rel_err(db, dy.double().sum(0)) < X3_TOL
    # the hi / lo split really carries the low bits: plain bf16 rounding of the operands would be ~4e-3
    assert rel_err(y, (x.bfloat16().double() @ w.bfloat16().double() + b.double()).relu()) > 1e-4 or M * K * N < 1000


def test_bf16x3_dual_a_and_strided_input():
    ops = _ops()
    from transformertts_amd import _lib
    X3 = _lib.TTSMI_BF16X3
    M, K1, K2, N = 300, 64, 192, 96
    x, x2, w, b = g(M, K1, seed=1), g(M, K2, seed=4), g(K1 + K2, N, seed=2), g(N, seed=3)
    y = ops.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), False, x2.to(DEV), dtype=X3)
    assert rel_err(y, torch.cat([x, x2], 1).double() @ w.double() + b.double()) < X3_TOL
    big = g(100, 192, seed=5).to(DEV)
    w2 = g(64, 32, seed=6)
    y2 = ops.linear_fwd(big[:, 64:128], w2.to(DEV), None, False, dtype=X3)
    assert rel_err(y2, big.cpu()[:, 64:128].double() @ w2.double()) < X3_TOL
    n = 160                                   # A = I with an asymmetric B: a transposed C write shows
    wa = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 97) - 3.0 * torch.arange(n)[None, :]
    assert rel_err(ops.linear_fwd(torch.eye(n).to(DEV), wa.to(DEV), None, False, dtype=X3), wa) < X3_TOL


@pytest.mark.parametrize('B,T,Cin,Cout,k', [(2, 50, 64, 64, 3), (3, 17, 32, 226, 3), (2, 40, 226, 8, 3), (1, 9, 4, 4, 5), (2, 30, 16, 24, 1),
                                            (4, 200, 256, 256, 3), (2, 13, 12, 20, 4)])
def test_bf16x3_conv1d_family(B, T, Cin, Cout, k):
    ops = _ops()
    from transformertts_amd import _lib
    X3 = _lib.TTSMI_BF16X3
    x, w, b = g(B, T, Cin, seed=1), g(k, Cin, Cout, seed=2, scale=0.2), g(Cout, seed=3)
    xd, wd, bd = (t.double().requires_grad_() for t in (x, w, b))
    y = ops.conv1d_fwd(x.to(DEV), w.to(DEV), b.to(DEV), relu=True, dtype=X3)
    assert rel_err(y, fo.conv1d_same(xd, wd, bd).relu().detach()) < X3_TOL
    dy = g(B, T, Cout, seed=4)
    fo.conv1d_same(xd, wd, bd).backward(dy.double())
    assert rel_err(ops.conv1d_dgrad(dy.to(DEV), w.to(DEV), dtype=X3), xd.grad) < X3_TOL
    dw, db = torch.empty_like(w, device=DEV), torch.empty(Cout, device=DEV)
    ops.conv1d_wgrad(x.to(DEV), dy.to(DEV), dw, db, dtype=X3)
    assert rel_err(dw, wd.grad) < X3_TOL
    assert rel_err(db, bd.grad) < X3_TOL


def test_ffn_and_convstack_autograd():
    ops = _ops()
    M, d, F = 300, 64, 256
    x, w1, b1, w2, b2 = g(M, d, seed=1), g(d, F, seed=2, scale=0.2), g(F, seed=3), g(F, d, seed=4, scale=0.2), g(d, seed=5)
    ts = [t.to(DEV).requires_grad_() for t in (x, w1, b1, w2, b2)]
    y = ops.FFNFn.apply(*ts, None, None, None, None)
    dy = g(M, d, seed=6)
    y.backward(dy.to(DEV))
    td = [t.double().requires_grad_() for t in (x, w1, b1, w2, b2)]
    yd = (td[0] @ td[1] + td[2]).relu() @ td[3] + td[4]
    yd.backward(dy.double())
    assert rel_err(y.detach(), yd.detach()) < 2e-6
    for a, b in zip(ts, td):
        assert rel_err(a.grad, b.grad) < 5e-6
    # conv stack (CNNResNorm convs): conv -> relu -> conv
    B, T, C, Fh = 2, 40, 32, 96
    x, w0, b0, w1, b1 = g(B, T, C, seed=1), g(3, C, Fh, seed=2, scale=0.2), g(Fh, seed=3), g(3, Fh, C, seed=4, scale=0.2), g(C, seed=5)
    ts = [t.to(DEV).requires_grad_() for t in (x, w0, b0, w1, b1)]
    y = ops.ConvStackFn.apply(ts[0], 2, None, *ts[1:])
    dy = g(B, T, C, seed=6)
    y.backward(dy.to(DEV))
    td = [t.double().requires_grad_() for t in (x, w0, b0, w1, b1)]
    yd = fo.conv1d_same(fo.conv1d_same(td[0], td[1], td[2]).relu(), td[3], td[4])
    yd.backward(dy.double())
    assert rel_err(y.detach(), yd.detach()) < 2e-6
    for a, b in zip(ts, td):
        assert rel_err(a.grad, b.grad) < 5e-6


@pytest.mark.parametrize('k', [3, 5])
def test_bf16_conv_stack_autograd(monkeypatch, k):
    """ConvStackFn with bf16 shadows (the conv blocks of the reference-default architecture, model/layers.py:30-38) on both
    routes - 'plain': a GEMM with overlapping A rows over a zero-margin bf16 layout (lda = C, K = k C), weight gradients
    per tap on shifted views; 'window': the implicit-GEMM kernel on fp32 activations - against the fp64 conv of the
    bf16-rounded operands.  Three sequences, so two interior sequence boundaries: a row computed across one of them
    must contribute nothing, forward or backward.  The hidden bias is +-8 per channel, so that no pre-activation sits
    near ReLU's zero (with 135 rows a single bf16-induced flip moves a bias gradient by several per cent)."""
    ops = _ops()
    B, T, C, Fh = 3, 45, 64, 128
    x, w0, w1, b1 = g(B, T, C, seed=1), g(k, C, Fh, seed=2, scale=0.1), g(k, Fh, C, seed=4, scale=0.1), g(C, seed=5)
    b0 = 8.0 * (1 - 2 * (torch.arange(Fh) % 2).float())
    dy = g(B, T, C, seed=6)
    r = lambda t: t.to(torch.bfloat16).double()
    td = [r(x).requires_grad_(), r(w0).requires_grad_(), b0.double().requires_grad_(), r(w1).requires_grad_(),
          b1.double().requires_grad_()]
    yd = fo.conv1d_same(fo.conv1d_same(td[0], td[1], td[2]).relu(), td[3], td[4])
    yd.backward(dy.double())
    runs = {}
    for plain in (True, False):
        monkeypatch.setattr(ops, '_CONV_PLAIN', plain)
        ts = [t.to(DEV).requires_grad_() for t in (x, w0, b0, w1, b1)]
        shs = (ops.make_shadow(ts[1]), ops.make_shadow(ts[3]))
        y = ops.ConvStackFn.apply(ts[0], 2, shs, *ts[1:])
        y.backward(dy.to(DEV))
        ops.wgrad_join()
        torch.cuda.synchronize()
        # bf16 hidden activation and bf16 gradient operands: 2^-9 relative per element
        assert rel_err(y.detach(), yd.detach()) < 6e-3, plain
        for a, b in zip(ts, td):
            assert rel_err(a.grad, b.grad) < 8e-3, plain
        runs[plain] = [y.detach()] + [t.grad for t in ts]
    for a, b in zip(runs[True], runs[False]):        # same operand rounding on both routes: fp32 summation order apart
        assert rel_err(a, b) < 2e-3


# ------------------------------------------------------------------------------------ layernorm
@pytest.mark.parametrize('M,C', [(37, 64), (100, 256), (9, 226), (50, 384), (20, 1024), (5, 30), (3, 1536)])
@pytest.mark.parametrize('mode', ['plain', 'res_mask', 'pe', 'relu_in'])
def test_add_layernorm(M, C, mode):
    ops = _ops()
    T = 7
    x, res, gam, bet = g(M, C, seed=1), g(M, C, seed=2), g(C, seed=3) * 0.3 + 1, g(C, seed=4)
    pe, ps = g(T, C, seed=5), torch.tensor(0.7)
    pad = (torch.arange(M) % 5 == 0).to(torch.uint8)
    if mode == 'relu_in':
        x = x.relu()
    kw = {}
    xd, rd, gd, bd, psd = (t.double().requires_grad_() for t in (x, res, gam, bet, ps))
    if mode == 'plain':
        want = fo.layer_norm(xd, gd, bd)
    elif mode == 'relu_in':
        pre = g(M, C, seed=1).double().requires_grad_()
        want = fo.layer_norm(pre.relu(), gd, bd)
        kw = dict(relu_in=True)
    elif mode == 'res_mask':
        want = fo.layer_norm(xd + rd, gd, bd) * (1 - pad.double())[:, None]
        kw = dict(row_pad=pad.to(DEV))
    else:
        want = fo.layer_norm(xd, gd, bd) + psd * pe.double()[torch.arange(M) % T]
        kw = dict(pe=pe.to(DEV), T=T)
    xg, rg, gg, bg, psg = (t.to(DEV).requires_grad_() for t in (x, res, gam, bet, ps))
    y = ops.add_layernorm(xg, rg if mode == 'res_mask' else None, gg, bg,
                          pe_scale=psg if mode == 'pe' else None, **kw)
    assert rel_err(y.detach(), want.detach()) < 3e-6
    dy = g(M, C, seed=9)
    y.backward(dy.to(DEV))
    want.backward(dy.double())
    assert rel_err(gg.grad, gd.grad) < 1e-5 and rel_err(bg.grad, bd.grad) < 1e-5
    if mode == 'relu_in':
        assert rel_err(xg.grad, pre.grad) < 1e-5
    else:
        assert rel_err(xg.grad, xd.grad) < 1e-5
    if mode == 'res_mask':
        assert rel_err(rg.grad, rd.grad) < 1e-5
    if mode == 'pe':
        assert rel_err(psg.grad, psd.grad) < 1e-5


def test_add_layernorm_dropout_statistics_and_determinism():
    ops = _ops()
    M, C, p = 4000, 256, 0.1
    x, res = g(M, C, seed=1).to(DEV), g(M, C, seed=2).to(DEV)
    gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    drop = ops.DropCtx(seed=123)
    y0 = ops.add_layernorm(x, None, gam, bet)
    y1 = ops.add_layernorm(x, None, gam, bet, p_out=p, site_out=3, drop=drop)
    y2 = ops.add_layernorm(x, None, gam, bet, p_out=p, site_out=3, drop=drop)
    y3 = ops.add_layernorm(x, None, gam, bet, p_out=p, site_out=4, drop=drop)
    assert torch.equal(y1, y2)                      # same (seed, site) -> same mask
    assert not torch.equal(y1, y3)                  # different site -> different mask
    kept = (y1 != 0)
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - p) < 4 * math.sqrt(p * (1 - p) / (M * C))
    assert torch.allclose(y1[kept], (y0 / (1 - p))[kept], rtol=1e-6, atol=1e-7)
    # mask is spatially uncorrelated enough: per-row and per-column drop rates concentrate around p
    assert abs((~kept).float().mean(0).std().item() - math.sqrt(p * (1 - p) / M)) < 2e-3
    # backward regenerates the same masks (p_in on the x branch, res untouched)
    xg, rg = x.clone().requires_grad_(), res.clone().requires_grad_()
    ya = ops.add_layernorm(xg, rg, gam, bet, p_in=p, site_in=7, drop=drop)
    (ya * g(M, C, seed=8).to(DEV)).sum().backward()
    zero_in_x = (xg.grad == 0).float().mean().item()
    assert abs(zero_in_x - p) < 0.01 and (rg.grad == 0).float().mean().item() < 1e-3
    # a device step counter changes the stream without changing kernel arguments
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    d2 = ops.DropCtx(seed=123, step_dev=step)
    a = ops.add_layernorm(x, None, gam, bet, p_out=p, site_out=3, drop=d2)
    ops.step_increment(step)
    b = ops.add_layernorm(x, None, gam, bet, p_out=p, site_out=3, drop=d2)
    assert torch.equal(a, y1) and not torch.equal(a, b)


# ------------------------------------------------------------------------------------ attention
def _attn_ref(qkv, pad, B, H, T, dh):
    d = H * dh
    q, k, v = qkv.double().reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
    mask = pad.double()[:, None, None, :]
    # the reference adds mask*-1e9 in fp32; emulate the fp32 absorption of the logit
    logits32 = (q @ k.transpose(-1, -2) / math.sqrt(dh)).float() + (mask * -1e9).float()
    w = torch.softmax(logits32.double(), -1)
    ctx = (w @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    return ctx, w


@pytest.mark.parametrize('B,H,T,dh', [(2, 2, 50, 32), (3, 4, 200, 64), (1, 1, 1, 32), (2, 4, 333, 64),
                                      (2, 2, 129, 64), (1, 4, 900, 64), (2, 2, 150, 192), (1, 2, 70, 96)])
def test_attention_fwd_bwd_weights(B, H, T, dh):
    ops = _ops()
    d = H * dh
    qkv = g(B * T, 3 * d, seed=1)
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 1)) for i in range(B - 1)])
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    if T > 10:
        pad[0, 3] = 1                              # an interior padded key (token id 0 mid-sequence)
    padg = pad.to(DEV)
    klen = torch.zeros(B, dtype=torch.int32)       # klen = 1 + last unpadded index
    for i, p in enumerate(pad):
        nz = (p == 0).nonzero()
        klen[i] = T if len(nz) == 0 else int(nz.max()) + 1
    qg = qkv.to(DEV).requires_grad_()
    ctx, lse = ops.AttentionFn.apply(qg, padg, klen.to(DEV), B, H, T, dh, 0.0, None, 0)
    qd = qkv.double().requires_grad_()
    want, wts = _attn_ref(qd, pad, B, H, T, dh)
    assert rel_err(ctx.detach(), want.detach()) < 5e-6
    w_gpu = ops.attention_weights(qg.detach(), padg, lse, B, H, T, dh)
    assert rel_err(w_gpu, wts.detach()) < 5e-6
    dctx = g(B * T, d, seed=2)
    ctx.backward(dctx.to(DEV))
    want.backward(dctx.double())
    assert rel_err(qg.grad, qd.grad) < 2e-5


def test_attention_all_keys_padded_is_uniform():
    ops = _ops()
    B, H, T, dh = 1, 2, 40, 32
    qkv = g(B * T, 3 * H * dh, seed=1)
    pad = torch.ones(B, T, dtype=torch.uint8)
    padg, klen = ops.length_pad_mask(torch.zeros(B, dtype=torch.int32, device=DEV), T)
    assert torch.equal(padg.cpu(), pad) and int(klen[0]) == T
    ctx, lse = ops.AttentionFn.apply(qkv.to(DEV), padg, klen, B, H, T, dh, 0.0, None, 0)
    want, _ = _attn_ref(qkv, pad, B, H, T, dh)
    assert rel_err(ctx, want) < 1e-5


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump at a late key tile (guide rule 26)."""
    ops = _ops()
    B, H, T, dh = 1, 1, 300, 64
    qkv = g(B * T, 3 * dh, seed=3)
    qkv[:, :dh] *= 0.1
    qkv[5, :dh] = 3.0
    qkv[250, dh:2 * dh] = 3.0                      # key 250 spikes against query 5
    pad = torch.zeros(B, T, dtype=torch.uint8)
    klen = torch.full((B,), T, dtype=torch.int32)
    ctx, lse = ops.AttentionFn.apply(qkv.to(DEV), pad.to(DEV), klen.to(DEV), B, H, T, dh, 0.0, None, 0)
    want, wts = _attn_ref(qkv, pad, B, H, T, dh)
    assert float(wts[0, 0, 5, 250]) > 0.99
    assert rel_err(ctx, want) < 5e-6


def test_attention_dropout_consistency():
    """weights kernel, forward and backward all regenerate the same dropout mask."""
    ops = _ops()
    B, H, T, dh, p = 2, 2, 150, 32, 0.1
    d = H * dh
    qkv = g(B * T, 3 * d, seed=1)
    pad = torch.zeros(B, T, dtype=torch.uint8)
    klen = torch.full((B,), T, dtype=torch.int32)
    drop = ops.DropCtx(seed=99)
    qg = qkv.to(DEV).requires_grad_()
    ctx, lse = ops.AttentionFn.apply(qg, pad.to(DEV), klen.to(DEV), B, H, T, dh, p, drop, 5)
    w = ops.attention_weights(qg.detach(), pad.to(DEV), lse, B, H, T, dh, p, drop, 5).cpu().double()
    frac = (w == 0).double().mean().item()
    assert abs(frac - p) < 0.005
    # rebuild ctx / grads on the CPU from the materialised dropped weights
    qd = qkv.double().requires_grad_()
    q, k, v = qd.reshape(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
    sm = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1)
    keep = (w != 0).double() / (1 - p)
    want = ((sm * keep) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    assert rel_err(ctx.detach(), want.detach()) < 1e-5
    dctx = g(B * T, d, seed=2)
    ctx.backward(dctx.to(DEV))
    want.backward(dctx.double())
    assert rel_err(qg.grad, qd.grad) < 3e-5


# ------------------------------------------------------------------------------------ small ops
def test_masks_embedding_pitch_rowdot():
    ops = _ops()
    B, T, V, C = 3, 37, 127, 64
    tok = torch.randint(1, V, (B, T), generator=torch.Generator().manual_seed(0)).int()
    tok[0, 20:] = 0
    tok[1, 5] = 0
    tok[2, :] = 0
    pad, klen = ops.token_pad_mask(tok.to(DEV))
    assert torch.equal(pad.cpu(), (tok == 0).to(torch.uint8))
    assert klen.cpu().tolist() == [20, T, T]
    pad2, klen2 = ops.length_pad_mask(torch.tensor([5, 0, T + 3], dtype=torch.int32, device=DEV), T)
    assert klen2.cpu().tolist() == [5, T, T]
    assert pad2.cpu()[0].tolist() == [0] * 5 + [1] * (T - 5)
    table = g(V, C, seed=1)
    tg = table.to(DEV).requires_grad_()
    e = ops.EmbeddingFn.apply(tok.to(DEV), tg, None)
    assert torch.equal(e.detach().cpu(), table[tok.long()])
    dy = g(B, T, C, seed=2)
    e.backward(dy.to(DEV))
    td = table.double().requires_grad_()
    td[tok.long()].backward(dy.double())
    assert rel_err(tg.grad, td.grad) < 1e-6
    # benchmark-size token matrix: several scan chunks, one id (padding) with thousands of positions, unused ids
    tok_l = torch.randint(1, 40, (32, 200), generator=torch.Generator().manual_seed(5)).int()
    tok_l[:, 60:] = 0
    tl = g(V, 256, seed=12).to(DEV).requires_grad_()
    dyl = g(32, 200, 256, seed=13)
    ops.EmbeddingFn.apply(tok_l.to(DEV), tl, None).backward(dyl.to(DEV))
    want = torch.zeros(V, 256, dtype=torch.float64).index_add_(0, tok_l.reshape(-1).long(), dyl.reshape(-1, 256).double())
    assert rel_err(tl.grad, want) < 2e-5 and float(tl.grad[40:].abs().max()) == 0.0      # fp32 running sums of 4 480 rows
    # pitch embed
    x, p, w, b = g(B * T, C, seed=3), g(B * T, seed=4), g(C, seed=5), g(C, seed=6)
    ts = [t.to(DEV).requires_grad_() for t in (x, p, w, b)]
    y = ops.PitchEmbedFn.apply(*ts, None, None)
    tdl = [t.double().requires_grad_() for t in (x, p, w, b)]
    yd = tdl[0] + (tdl[1][:, None] * tdl[2][None, :] + tdl[3]).relu()
    dy = g(B * T, C, seed=7)
    y.backward(dy.to(DEV))
    yd.backward(dy.double())
    assert rel_err(y.detach(), yd.detach()) < 1e-6
    for a, bb in zip(ts, tdl):
        assert rel_err(a.grad, bb.grad) < 1e-5
    # rowdot head
    for relu in (False, True):
        x, w, b = g(B, T, 226, seed=8), g(226, 1, seed=9), g(1, seed=10)
        ts = [t.to(DEV).requires_grad_() for t in (x, w, b)]
        y = ops.RowDotFn.apply(*ts, None, None, pad, relu)
        tdl = [t.double().requires_grad_() for t in (x, w, b)]
        yd = tdl[0] @ tdl[1] + tdl[2]
        if relu:
            yd = yd.relu()
        yd = yd * (1 - pad.cpu().double())[..., None]
        dy = g(B, T, 1, seed=11)
        y.backward(dy.to(DEV))
        yd.backward(dy.double())
        assert rel_err(y.detach(), yd.detach()) < 1e-6
        for a, bb in zip(ts, tdl):
            assert rel_err(a.grad, bb.grad) < 1e-5


# ------------------------------------------------------------------------------------ length regulator
@pytest.mark.parametrize('seed', range(4))
def test_lenreg_bit_exact(seed):
    ops = _ops()
    rng = np.random.default_rng(seed)
    B, Tp, C = 4, 300, 64
    dur = rng.choice([0., 0.5, 1.5, 2.5, 3.5, 1.0, 2.0, 7.49, 7.5, 0.49999997, 12.0], size=(B, Tp)).astype(np.float32)
    if seed == 0:
        dur[1] = 0
    if seed == 1:
        dur[2, 7] = -3.0                           # negative duration -> clamped to 0 (documented)
    x = rng.standard_normal((B, Tp, C)).astype(np.float32)
    d_or = np.maximum(dur, 0)[..., None]
    idx_w, len_w, out_len = fo.expand_indices_np(d_or)
    cap = max(out_len, 1)
    idx, cum, ln = ops.lenreg_index(torch.from_numpy(dur).to(DEV), cap)
    np.testing.assert_array_equal(ln.cpu().numpy(), len_w)
    np.testing.assert_array_equal(idx.cpu().numpy()[:, :out_len], idx_w)
    dims = np.rint(np.maximum(dur, 0)).astype(np.int32)
    np.testing.assert_array_equal(cum.cpu().numpy()[:, 1:], np.cumsum(dims, 1))
    xg = torch.from_numpy(x).to(DEV).requires_grad_()
    y = ops.LenRegFn.apply(xg, idx, cum)
    want = fo.expand_literal_np(x, d_or)
    np.testing.assert_array_equal(y.detach().cpu().numpy()[:, :out_len], want)
    # integer durations give the same table; truncating cap drops frames like the [:, :mel_len] slice
    idx_i, _, _ = ops.lenreg_index(torch.from_numpy(dims).to(DEV), cap)
    assert torch.equal(idx_i, idx)
    cap2 = max(out_len // 2, 1)
    idx_t, cum_t, _ = ops.lenreg_index(torch.from_numpy(dur).to(DEV), cap2)
    assert torch.equal(idx_t, idx[:, :cap2])
    # backward = segment sum
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32)).to(DEV)
    y.backward(dy)
    xd = torch.from_numpy(x).double().requires_grad_()
    fo.expand_torch(xd, torch.from_numpy(d_or)).backward(dy.cpu().double()[:, :out_len])
    assert rel_err(xg.grad, xd.grad) < 1e-6


def test_expand_docstring_example_on_gpu():
    ops = _ops()
    x = torch.tensor([[[0.54710746, 0.8943467], [0.7140938, 0.97968304], [0.5347662, 0.15213418]]])
    idx, cum, ln = ops.lenreg_index(torch.tensor([[1, 3, 2]], dtype=torch.int32, device=DEV), 6)
    xg = x.to(DEV).requires_grad_()
    yg = ops.LenRegFn.apply(xg, idx, cum)
    y = yg.detach().cpu()
    assert idx.cpu().tolist() == [[0, 1, 1, 1, 2, 2]] and int(ln[0]) == 6
    assert torch.equal(y, x[:, [0, 1, 1, 1, 2, 2]])
    yg.backward(torch.ones_like(yg))               # two channels: the scalar (C % 4 != 0) segment-sum path
    assert xg.grad.cpu().tolist() == [[[1., 1.], [3., 3.], [2., 2.]]]


# ------------------------------------------------------------------------------------ loss + optimiser
def test_l1_loss_and_adam():
    ops = _ops()
    pred, tgt = g(7, 33, 80, seed=1), g(7, 33, 80, seed=2)
    pred[0, 0, :5] = tgt[0, 0, :5]                 # exact zeros of |t - p|: sign(0) = 0
    pg = pred.to(DEV).requires_grad_()
    loss = ops.L1LossFn.apply(pg, tgt.to(DEV))
    (3.0 * loss).backward()
    pd = pred.double().requires_grad_()
    ld = (tgt.double() - pd).abs().mean()
    (3.0 * ld).backward()
    assert abs(float(loss) - float(ld)) / float(ld) < 1e-6
    assert rel_err(pg.grad, pd.grad) < 1e-6
    ti = torch.randint(0, 9, (4, 50, 1), generator=torch.Generator().manual_seed(3)).int()
    pr = g(4, 50, 1, seed=4)
    li = ops.L1LossFn.apply(pr.to(DEV), ti.to(DEV))
    assert abs(float(li) - float((ti.double() - pr.double()).abs().mean())) < 1e-6
    # weighted_sum_losses as one op: bit-identical to the three separate losses + the Python-level weighted sum
    from transformertts_amd.utils.losses import masked_mean_absolute_error as mae, weighted_sum_losses
    mel_p, mel_t = g(4, 50, 80, seed=20), g(4, 50, 80, seed=21)
    dur_p, pit_p, pit_t = g(4, 50, 1, seed=22), g(4, 50, 1, seed=23), g(4, 50, 1, seed=24)
    for unit_seed in (False, True):
        a = [t.to(DEV).requires_grad_() for t in (mel_p, dur_p, pit_p)]
        b = [t.to(DEV).requires_grad_() for t in (mel_p, dur_p, pit_p)]
        tg = (mel_t.to(DEV), ti.to(DEV), pit_t.to(DEV))
        total, vals = weighted_sum_losses(tg, a, [mae] * 3, [1., 1., 3.], unit_seed=unit_seed)
        sep = [ops.L1LossFn.apply(b[i], tg[i]) for i in range(3)]
        want = 0
        for c, l in zip([1., 1., 3.], sep):
            want = want + c * l
        assert [float(v) for v in vals] == [float(v) for v in sep] and float(total) == float(want)
        assert not any(v.requires_grad for v in vals) and total.requires_grad
        total.backward()
        want.backward()
        for x, y in zip(a, b):
            assert torch.equal(x.grad, y.grad)
    a = [t.to(DEV).requires_grad_() for t in (mel_p, dur_p, pit_p)]
    total, _ = weighted_sum_losses(tg, a, [mae] * 3, [1., 1., 3.])
    (2.0 * total).backward()                       # a seed other than 1 goes through the general path
    assert torch.equal(a[0].grad, 2.0 * b[0].grad) and torch.equal(a[2].grad, 2.0 * b[2].grad)
    # TF-form Adam, three steps, against the oracle's hand restatement
    n = 1000
    p0, grads = g(n, seed=5), [g(n, seed=6 + i) for i in range(3)]
    p, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], device=DEV)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    pw, mw, vw = p0.double().clone(), torch.zeros(n).double(), torch.zeros(n).double()
    for t, gr in enumerate(grads, start=1):
        ops.step_increment(step)
        ops.adam_tf(p, gr.to(DEV), m, v, lr, step)
        lr_t = 1e-3 * math.sqrt(1 - 0.98 ** t) / (1 - 0.9 ** t)
        mw = 0.9 * mw + 0.1 * gr.double()
        vw = 0.98 * vw + 0.02 * gr.double() ** 2
        pw = pw - lr_t * mw / (vw.sqrt() + 1e-9)
    assert int(step.item()) == 3
    assert rel_err(p, pw) < 1e-6


# ------------------------------------------------------------------------------------ bf16 GEMM path
def _bf(x):
    return x.to(torch.bfloat16).double()


@pytest.mark.parametrize('M,K,N', [(128, 256, 256), (333, 64, 200), (1000, 512, 80), (77, 1024, 130), (5, 8, 3)])
def test_hgemm_tn_matches_bf16_rounded_reference(M, K, N):
    ops = _ops()
    x, w, b = g(M, K, seed=1), g(K, N, seed=2), g(N, seed=3)
    sh = ops.make_shadow(w.to(DEV))
    assert torch.equal(sh.wb.cpu(), w.to(torch.bfloat16)) and torch.equal(sh.wt.cpu(), w.t().to(torch.bfloat16))
    y = ops.hgemm_tn(x.to(DEV), sh.wt, b.to(DEV), relu=True)
    want = (_bf(x) @ _bf(w) + b.double()).relu()
    assert rel_err(y, want) < 3e-6                       # only fp32 accumulation error remains
    dy, h = g(M, N, seed=4), g(M, K, seed=5)
    if N % 8 == 0:
        dx = ops.hgemm_tn(dy.to(DEV), sh.wb, relu_src=h.to(DEV))
        assert rel_err(dx, (_bf(dy) @ _bf(w).T) * (h > 0)) < 3e-6
    dw = torch.empty(K, N, device=DEV)
    db = torch.empty(N, device=DEV)
    ops.hgemm_wgrad(ops.cast_transpose_bf16(x.to(DEV)), ops.cast_transpose_bf16(dy.to(DEV)), dw, db, M)
    assert rel_err(dw, _bf(x).T @ _bf(dy)) < 3e-6
    assert rel_err(db, _bf(dy).sum(0)) < 3e-6
    # and the whole thing is a faithful bf16 approximation of the fp32 op
    assert rel_err(y, (x.double() @ w.double() + b.double()).relu()) < 2e-2


def test_hgemm_dual_a_and_cast_transpose_conv():
    ops = _ops()
    M, K1, K2, N = 300, 64, 192, 72
    x, x2, w = g(M, K1, seed=1), g(M, K2, seed=2), g(K1 + K2, N, seed=3)
    sh = ops.make_shadow(w.to(DEV))
    y = ops.hgemm_tn(x.to(DEV), sh.wt, None, False, x2.to(DEV))
    assert rel_err(y, torch.cat([_bf(x), _bf(x2)], 1) @ _bf(w)) < 3e-6
    # transposed im2col written by the cast kernel == unfold of the zero-padded sequence
    B, T, C, k = 3, 11, 8, 3
    xs = g(B, T, C, seed=4)
    xt = ops.cast_transpose_bf16(xs.reshape(B * T, C).to(DEV), taps=k, T=T, pad=1).cpu().float()
    assert xt.shape == (k * C, (B * T + 7) // 8 * 8)
    xp = torch.nn.functional.pad(xs.to(torch.bfloat16).float(), (0, 0, 1, 1))
    for j in range(k):
        want = xp[:, j:j + T, :].reshape(B * T, C).t()
        assert torch.equal(xt[j * C:(j + 1) * C, :B * T], want)
    assert float(xt[:, B * T:].abs().sum()) == 0.0


@pytest.mark.parametrize('Cout', [64, 226])
def test_bf16_conv_predictor_layer_autograd(Cout):
    ops = _ops()
    B, T, Cin, k = 2, 37, 64, 3
    x, w, b = g(B, T, Cin, seed=1), g(k, Cin, Cout, seed=2, scale=0.2), g(Cout, seed=3)
    ts = [t.to(DEV).requires_grad_() for t in (x, w, b)]
    sh = ops.make_shadow(ts[1])
    h = ops.ConvReluPreMaskedFn.apply(ts[0], ts[1], ts[2], None, None, sh)
    td = [_bf(x).requires_grad_(), _bf(w).requires_grad_(), b.double().requires_grad_()]
    hd = fo.conv1d_same(td[0], td[1], td[2]).relu()
    assert rel_err(h.detach(), hd.detach()) < 3e-6
    dh = g(B, T, Cout, seed=4) * (hd.detach() > 0).float()          # pre-masked upstream gradient
    h.backward(dh.to(DEV))
    # reference backward with bf16-rounded dh on the GEMM operands
    pre = fo.conv1d_same(td[0], td[1], td[2])
    pre.backward(_bf(dh))
    assert rel_err(ts[1].grad, td[1].grad) < 1e-5
    assert rel_err(ts[0].grad, td[0].grad) < (1e-5 if Cout % 8 == 0 else 2e-2)   # 226: fp32 dgrad of unrounded dh
    assert rel_err(ts[2].grad, _bf(dh).sum((0, 1))) < 1e-5


@pytest.mark.parametrize('B,H,T,dh', [(2, 2, 50, 32), (3, 4, 200, 64), (2, 4, 333, 64), (1, 4, 900, 64), (2, 2, 150, 192),
                                      (1, 2, 333, 192)])
def test_bf16_attention_fwd_bwd(B, H, T, dh):
    """TTSMI_BF16 attention vs the fp64 reference evaluated on bf16-rounded q/k/v: what remains is
    the bf16 rounding of P (and dS / dO), i.e. ~2^-9 relative per element, averaged by the sums."""
    ops = _ops()
    from transformertts_amd._lib import TTSMI_BF16
    d = H * dh
    qkv = g(B * T, 3 * d, seed=1).to(torch.bfloat16).float()           # exactly representable inputs
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 1)) for i in range(B - 1)])
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    if T > 10:
        pad[0, 3] = 1
    klen = torch.zeros(B, dtype=torch.int32)
    for i, p in enumerate(pad):
        nz = (p == 0).nonzero()
        klen[i] = T if len(nz) == 0 else int(nz.max()) + 1
    qg = qkv.to(DEV).requires_grad_()
    ctx, lse = ops.AttentionFn.apply(qg, pad.to(DEV), klen.to(DEV), B, H, T, dh, 0.0, None, 0, TTSMI_BF16)
    qd = qkv.double().requires_grad_()
    want, _ = _attn_ref(qd, pad, B, H, T, dh)
    assert rel_err(ctx.detach(), want.detach()) < 6e-3
    dctx = g(B * T, d, seed=2).to(torch.bfloat16).float()
    ctx.backward(dctx.to(DEV))
    want.backward(dctx.double())
    assert rel_err(qg.grad, qd.grad) < 2e-2
    # fp32 kernels on the same inputs agree with the bf16 kernels to bf16 accuracy
    q2 = qkv.to(DEV).requires_grad_()
    c2, _ = ops.AttentionFn.apply(q2, pad.to(DEV), klen.to(DEV), B, H, T, dh, 0.0, None, 0)
    assert rel_err(ctx.detach(), c2.detach()) < 6e-3


def test_bf16_attention_dropout_matches_fp32_mask():
    """Both precisions draw the SAME dropout mask (same hash, same element index)."""
    ops = _ops()
    from transformertts_amd._lib import TTSMI_BF16
    B, H, T, dh, p = 2, 2, 150, 32, 0.25
    qkv = g(B * T, 3 * H * dh, seed=1).to(torch.bfloat16).float().to(DEV)
    pad = torch.zeros(B, T, dtype=torch.uint8, device=DEV)
    klen = torch.full((B,), T, dtype=torch.int32, device=DEV)
    drop = ops.DropCtx(seed=7)
    a, _ = ops.AttentionFn.apply(qkv, pad, klen, B, H, T, dh, p, drop, 9, TTSMI_BF16)
    b, _ = ops.AttentionFn.apply(qkv, pad, klen, B, H, T, dh, p, drop, 9)
    c, _ = ops.AttentionFn.apply(qkv, pad, klen, B, H, T, dh, p, drop, 10)
    assert rel_err(a, b) < 1e-2 and rel_err(c, b) > 5e-2


@pytest.mark.parametrize('B,H,T,dh', [(2, 2, 150, 32), (3, 4, 200, 64), (2, 4, 900, 64), (1, 2, 33, 64), (2, 2, 100, 192)])
def test_attention_keep_bit_table_equals_the_hashed_dropout(B, H, T, dh):
    """ttsmi_attention_dropmask + the *_masked kernels (bf16 I/O) make the same keep decisions as the kernels that
    hash in their inner loops: forward context, log-sum-exp and dqkv agree to fp32 rounding order (the 1/keep
    factor is applied at a different point), with padded keys, ragged klen and T % 32 != 0."""
    ops = _ops()
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream, check
    l = _lib.lib()
    d, pdrop = H * dh, 0.2
    qkv = (g(B * T, 3 * d, seed=1) * 0.7).to(DEV).to(torch.bfloat16)
    dctx = (g(B * T, d, seed=2) * 0.3).to(DEV).to(torch.bfloat16)
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 1)) for i in range(B - 1)])
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    if T > 10:
        pad[0, 3] = 1
    klen = torch.tensor([T if not (p == 0).any() else int((p == 0).nonzero().max()) + 1 for p in pad], dtype=torch.int32)
    pad, klen = pad.to(DEV), klen.to(DEV)
    step = torch.full((1,), 5, dtype=torch.int64, device=DEV)
    drop = ops.DropCtx(seed=11, step_dev=step)
    site = 4
    ws = torch.empty(int(l.ttsmi_attention_bwd_ws_bytes(B, H, T, dh)), dtype=torch.uint8, device=DEV)
    outs = []
    for masked in (False, True):
        ctx = torch.empty(B * T, d, device=DEV, dtype=torch.bfloat16)
        lse = torch.empty(B, H, T, device=DEV)
        dqkv = torch.empty_like(qkv)
        if masked:
            m = ops.attention_dropmask(B, H, T, pdrop, drop, site, DEV)
            check(l.ttsmi_attention_fwd_masked(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, pdrop, _p(m),
                                               _lib.TTSMI_BF16_IO, _stream()))
            check(l.ttsmi_attention_bwd_masked(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(dctx), _p(lse), _p(dqkv), B, H, T,
                                               dh, pdrop, _p(m), _p(ws), ws.numel(), _lib.TTSMI_BF16_IO, _stream()))
        else:
            check(l.ttsmi_attention_fwd(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, pdrop, drop.seed,
                                        _p(step), site, _lib.TTSMI_BF16_IO, _stream()))
            check(l.ttsmi_attention_bwd(_p(qkv), _p(pad), _p(klen), _p(ctx), _p(dctx), _p(lse), _p(dqkv), B, H, T, dh,
                                        pdrop, drop.seed, _p(step), site, _p(ws), ws.numel(), _lib.TTSMI_BF16_IO, _stream()))
        torch.cuda.synchronize()
        outs.append((ctx.float().cpu(), lse.cpu(), dqkv.float().cpu()))
    (c0, l0, g0), (c1, l1, g1) = outs
    live = (torch.arange(T)[None, :] < lens[:, None]).reshape(-1)          # padded query rows are compared too (finite)
    assert torch.isfinite(c1).all() and torch.isfinite(g1).all()
    assert torch.equal(l0, l1)                                             # the softmax statistics do not see dropout
    assert rel_err(c1, c0) < 1e-2 and rel_err(g1, g0) < 1e-2               # one bf16 ulp where the scaling order differs
    # a different decision anywhere would move an element by O(1) of its value: the mean error stays at rounding level
    assert float((c1 - c0).abs().mean()) < 2e-3 * float(c0.abs().mean())
    assert float((g1 - g0).abs().mean()) < 2e-3 * float(g0.abs().mean())
    assert live.any()


def test_keep_bit_tables_of_a_stack_in_one_launch_equal_the_single_calls():
    """ttsmi_attention_dropmask_stack (the train step's form: every layer of a stack, one launch) writes the same bits as one
    ttsmi_attention_dropmask call per layer, T % 32 != 0 included."""
    import ctypes
    ops = _ops()
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream, check
    l = _lib.lib()
    B, H, T, pdrop, seed, n = 3, 4, 173, 0.1, 991, 5
    step = torch.full((1,), 7, dtype=torch.int64, device=DEV)
    nb = int(l.ttsmi_attention_dropmask_bytes(B, H, T))
    sites = [3, 9, 4, 100, 17]
    single = [torch.zeros(nb, dtype=torch.uint8, device=DEV) for _ in range(n)]
    stack = [torch.full((nb + 64,), 0x5A, dtype=torch.uint8, device=DEV) for _ in range(n)]
    for m, s_ in zip(single, sites):
        check(l.ttsmi_attention_dropmask(_p(m), B, H, T, pdrop, seed, _p(step), s_, _stream()))
    ptrs = (ctypes.c_void_p * n)(*[m.data_ptr() for m in stack])
    csites = (ctypes.c_uint32 * n)(*sites)
    check(l.ttsmi_attention_dropmask_stack(ctypes.addressof(ptrs), ctypes.addressof(csites), n, B, H, T, pdrop, seed, _p(step), _stream()))
    torch.cuda.synchronize()
    for a, b in zip(single, stack):
        assert torch.equal(a, b[:nb]) and bool((b[nb:] == 0x5A).all())
    assert not torch.equal(single[0], single[1])


@pytest.mark.parametrize('B,H,T,dh', [(1, 4, 2304, 64), (1, 4, 400, 64), (2, 2, 333, 32), (1, 2, 700, 192), (3, 4, 100, 64),
                                      (32, 4, 900, 64)])
def test_split_key_attention_forward_equals_the_plain_forward(B, H, T, dh):
    """ttsmi_attention_fwd_splitkeys (inference, small launches: keys split over extra workgroups + combine) against
    ttsmi_attention_fwd on the same bf16 qkv: contexts within the bf16 rounding of the partial contexts, log-sum-exp to
    fp32 rounding; ragged klen (one sequence ends inside the first split, so later splits are EMPTY for it), a padded
    key in the middle, T not a multiple of the tile.  The last case is big enough that no split is made."""
    ops = _ops()
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream, check
    l = _lib.lib()
    d = H * dh
    qkv = (g(B * T, 3 * d, seed=1) * 0.7).to(DEV).to(torch.bfloat16)
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 7)) for i in range(B - 1)])
    if B == 1:
        lens[0] = T - 5
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    pad[0, 3] = 1
    klen = torch.tensor([int((p == 0).nonzero().max()) + 1 for p in pad], dtype=torch.int32)
    pad, klen = pad.to(DEV), klen.to(DEV)
    need = int(l.ttsmi_attention_fwd_splitkeys_ws_bytes(B, H, T, dh))
    assert (need == 0) == (B == 32 or T <= 128)             # one staged pair of key tiles is not split either
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
    c0 = torch.empty(B * T, d, device=DEV, dtype=torch.bfloat16)
    c1 = torch.full_like(c0, float('nan'))
    l0 = torch.empty(B, H, T, device=DEV)
    l1 = torch.full_like(l0, float('nan'))
    check(l.ttsmi_attention_fwd(_p(qkv), _p(pad), _p(klen), _p(c0), _p(l0), B, H, T, dh, 0.0, 0, None, 0,
                                _lib.TTSMI_BF16_IO, _stream()))
    check(l.ttsmi_attention_fwd_splitkeys(_p(qkv), _p(pad), _p(klen), _p(c1), _p(l1), B, H, T, dh, _p(ws), ws.numel(),
                                          _stream()))
    torch.cuda.synchronize()
    assert torch.isfinite(c1.float()).all() and torch.isfinite(l1).all()
    assert rel_err(l1, l0) < 1e-6
    assert rel_err(c1.float(), c0.float()) < (1e-2 if need else 1e-9)
    assert float((c1.float() - c0.float()).abs().mean()) < 3e-3 * float(c0.float().abs().mean()) + 1e-12
    if need:
        with pytest.raises(_lib.TtsmiError):
            check(l.ttsmi_attention_fwd_splitkeys(_p(qkv), _p(pad), _p(klen), _p(c1), _p(l1), B, H, T, dh, _p(ws), 16,
                                                  _stream()))


@pytest.mark.parametrize('M,K,N', [(1000, 256, 128), (333, 64, 200), (28800 // 8, 1024, 256), (77, 100, 60)])
def test_hgemm_wgrad_rows(M, K, N):
    ops = _ops()
    x, dy = g(M, K, seed=1), g(M, N, seed=2)
    dw, db = torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
    ops.hgemm_wgrad_rows(x.to(DEV), dy.to(DEV), dw, db)
    assert rel_err(dw, _bf(x).T @ _bf(dy)) < 3e-6
    assert rel_err(db, _bf(dy).sum(0)) < 3e-6


def test_hgemm_wgrad_rows_conv():
    ops = _ops()
    B, T, Cin, Cout, k = 3, 41, 128, 72, 3
    x, dy = g(B, T, Cin, seed=1), g(B, T, Cout, seed=2)
    dw, db = torch.empty(k * Cin, Cout, device=DEV), torch.empty(Cout, device=DEV)
    ops.hgemm_wgrad_rows(x.reshape(B * T, Cin).to(DEV), dy.reshape(B * T, Cout).to(DEV), dw, db, conv=(k, T, Cin, 1))
    xd, wd = _bf(x), torch.zeros(k, Cin, Cout, dtype=torch.float64, requires_grad=True)
    fo.conv1d_same(xd, wd, torch.zeros(Cout, dtype=torch.float64)).backward(_bf(dy))
    assert rel_err(dw.reshape(k, Cin, Cout), wd.grad) < 3e-6


def test_hgemm_bf16_outputs_masks_and_sources():
    """FFN-internal tensors of the TTSMI_BF16 path live in bf16: GEMM output, ReLU mask, A operand and
    both wgrad sources."""
    ops = _ops()
    M, d, F = 500, 64, 256
    a, w1, b1, w2 = g(M, d, seed=1), g(d, F, seed=2, scale=0.2), g(F, seed=3), g(F, d, seed=4, scale=0.2)
    s1, s2 = ops.make_shadow(w1.to(DEV)), ops.make_shadow(w2.to(DEV))
    h1 = ops.hgemm_tn(a.to(DEV), s1.wt, b1.to(DEV), relu=True, out_bf16=True)
    assert h1.dtype == torch.bfloat16
    want_h1 = (_bf(a) @ _bf(w1) + b1.double()).relu()
    assert rel_err(h1.float(), want_h1) < 5e-3                      # one bf16 rounding of the output
    assert torch.equal(h1.cpu() > 0, want_h1.to(torch.bfloat16) > 0) or rel_err(h1.float(), want_h1) < 5e-3
    f = ops.hgemm_tn(h1, s2.wt)                                       # bf16 A operand
    assert rel_err(f, h1.double().cpu() @ _bf(w2)) < 3e-6
    df = g(M, d, seed=5)
    dh1 = ops.hgemm_tn(df.to(DEV), s2.wb, relu_src=h1, out_bf16=True)  # bf16 mask + bf16 out
    want_dh1 = (_bf(df) @ _bf(w2).T) * (h1.cpu() > 0)
    assert rel_err(dh1.float(), want_dh1) < 5e-3
    dw2, db2 = torch.empty(F, d, device=DEV), torch.empty(d, device=DEV)
    ops.hgemm_wgrad_rows(h1, df.to(DEV), dw2, db2)                    # bf16 x, fp32 dy
    assert rel_err(dw2, h1.double().cpu().T @ _bf(df)) < 3e-6
    dw1, db1 = torch.empty(d, F, device=DEV), torch.empty(F, device=DEV)
    ops.hgemm_wgrad_rows(a.to(DEV), dh1, dw1, db1)                    # fp32 x, bf16 dy
    assert rel_err(dw1, _bf(a).T @ dh1.double().cpu()) < 3e-6
    assert rel_err(db1, dh1.double().cpu().sum(0)) < 3e-6
    da = g(M, d, seed=6).to(DEV)
    da0 = da.clone()
    ops.hgemm_tn(dh1, s1.wb, out=da, accumulate=True)                 # bf16 A + accumulate
    assert rel_err(da, da0.double().cpu() + dh1.double().cpu() @ _bf(w1).T) < 3e-6


# (M >= 2048 takes the 128-row LDS-DMA kernel with its prefetched epilogue operands; below, the 64-row kernel)
@pytest.mark.parametrize('M,K,dual,pdrop', [(300, 512, True, 0.0), (1000, 1024, False, 0.15), (64, 256, False, 0.15), (129, 512, True, 0.15),
                                            (2100, 512, True, 0.15), (4000, 1024, False, 0.0), (3001, 768, False, 0.1)])
def test_hgemm_with_fused_layernorm_matches_gemm_then_layernorm(M, K, dual, pdrop):
    """ttsmi_hgemm_ln_fwd == hgemm_tn followed by add_layernorm (same dropout decisions, row mask, bf16 copy), and its
    x^ / rstd feed ttsmi_layernorm_bwd_xhat + ttsmi_hgemm_ln_bwd to the same gradients as the standalone LayerNorm
    backward (up to the bf16 rounding of x^)."""
    ops = _ops()
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream, check
    l = _lib.lib()
    N = 256
    a = g(M, K, seed=1).to(DEV).to(torch.bfloat16)
    w = (g(K, N, seed=2, scale=0.05)).to(DEV)
    sh = ops.make_shadow(w)
    bias, gam, bet = g(N, seed=3).to(DEV), (1 + 0.1 * g(N, seed=4)).to(DEV), (0.1 * g(N, seed=5)).to(DEV)
    res = g(M, N, seed=6).to(DEV)
    pad = (torch.arange(M) % 7 == 3).to(torch.uint8).to(DEV)
    step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    drop = ops.DropCtx(seed=21, step_dev=step)
    site = 5
    a1, a2 = (a[:, :K // 2].contiguous(), a[:, K // 2:].contiguous()) if dual else (a, None)
    # reference: unfused kernels
    o = ops.hgemm_tn(a1, sh.wt, bias, False, a2)
    y0, y0h, mean0, rstd0 = ops._ln_fwd(o, res, gam, bet, pad, pdrop, site, drop, True)
    # fused
    y = torch.empty(M, N, device=DEV)
    yh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    xh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    rstd = torch.empty(M, device=DEV)
    check(l.ttsmi_hgemm_ln_fwd(_p(a1), a1.stride(0), _p(a2), 0 if a2 is None else a2.stride(0), a1.shape[1] if dual else 0,
                               _p(sh.wt), sh.wt.stride(0), _p(bias), _p(res), _p(gam), _p(bet), _p(pad), pdrop, site,
                               drop.seed, _p(step), 1e-6, _p(y), _p(yh), _p(xh), _p(rstd), M, N, K, _stream()))
    torch.cuda.synchronize()
    assert rel_err(y, y0) < 2e-5 and rel_err(rstd, rstd0) < 2e-5
    assert rel_err(yh.float(), y0h.float()) < 1e-2
    live = (pad == 0)
    xhat_ref = ((y0 - bet) / gam)[live]
    assert rel_err(xh.float()[live], xhat_ref) < 1e-2
    # ---- backward of the same LayerNorm, standalone x^ form vs the classic kernel
    dy = g(M, N, seed=7).to(DEV)
    dg0, db0 = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dx0, dres0 = ops._ln_bwd(dy, o, res, gam, mean0, rstd0, pad, pdrop, site, drop, dg0, db0, dx_bf16=True)
    dxb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    dres = torch.empty(M, N, device=DEV)
    nw = int(l.ttsmi_layernorm_bwd_xhat_nparts(M))
    ws = torch.empty(int(l.ttsmi_layernorm_partials_bytes(nw, N)), dtype=torch.uint8, device=DEV)
    check(l.ttsmi_layernorm_bwd_xhat(_p(dy), _p(xh), _p(rstd), _p(gam), _p(pad), pdrop, site, drop.seed, _p(step), _p(dxb),
                                     _p(dres), _p(ws), ws.numel(), M, N, _stream()))
    torch.cuda.synchronize()
    assert rel_err(dres, dres0) < 1.5e-2 and rel_err(dxb.float(), dx0.float()) < 2e-2
    # parameter gradients from the partial rows the kernel left
    dg, db = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    with ops.ln_param_batch():
        ops._ln_defer(ws, dg, db, None, M, N, nw)
    torch.cuda.synchronize()
    assert rel_err(db, db0) < 1e-4 and rel_err(dg, dg0) < 1.5e-2
    # ---- GEMM + LayerNorm backward fused: dy = dy_part + a_b . w_b^T
    Kb = 1024
    ab = g(M, Kb, seed=8, scale=0.3).to(DEV).to(torch.bfloat16)
    wb = (g(N, Kb, seed=9, scale=0.05)).to(DEV)                    # "W as stored" [k_in = 256][n_out = Kb]: dgrad operand
    shb = ops.make_shadow(wb)
    part = g(M, N, seed=10).to(DEV)
    full = part.clone()
    ops.hgemm_tn(ab, shb.wb, out=full, accumulate=True)
    dg1, db1 = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    dx1, dres1 = ops._ln_bwd(full, o, res, gam, mean0, rstd0, pad, pdrop, site, drop, dg1, db1, dx_bf16=True)
    nw = int(l.ttsmi_hgemm_ln_bwd_nparts(M))
    ws = torch.empty(int(l.ttsmi_layernorm_partials_bytes(nw, N)), dtype=torch.uint8, device=DEV)
    check(l.ttsmi_hgemm_ln_bwd(_p(ab), ab.stride(0), _p(shb.wb), shb.wb.stride(0), _p(part), _p(xh), _p(rstd), _p(gam), _p(pad),
                               pdrop, site, drop.seed, _p(step), _p(dxb), _p(dres), _p(ws), ws.numel(), M, N, Kb, _stream()))
    dg, db = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    with ops.ln_param_batch():
        ops._ln_defer(ws, dg, db, None, M, N, nw)
    torch.cuda.synchronize()
    assert rel_err(dres, dres1) < 1.5e-2 and rel_err(dxb.float(), dx1.float()) < 2e-2
    assert rel_err(db, db1) < 1e-4 and rel_err(dg, dg1) < 1.5e-2


@pytest.mark.parametrize('M,N,relu,out_bf16', [(1000, 768, False, True), (333, 1024, True, True), (581, 256, False, False),
                                               (64, 80, True, False), (28800, 1024, True, True), (4100, 264, False, True),
                                               (20000, 768, False, True), (16500, 256, True, True)])
def test_weight_stationary_k256_gemm(M, N, relu, out_bf16, monkeypatch):
    """gemm_k256.hip (reached through ttsmi_hgemm_tn for K = 256 projections): row / column tails, every epilogue,
    against the bf16-rounded fp64 product; run in a subprocess-free way by forcing the route with TTSMI_HGEMM_K256=1
    before the library reads it (first call in this process decides: the test asserts the route it got)."""
    ops = _ops()
    from transformertts_amd import _lib
    K = 256
    x, w, b = g(M, K, seed=1), g(K, N, seed=2, scale=0.1), g(N, seed=3)
    sh = ops.make_shadow(w.to(DEV))
    xa = x.to(DEV).to(torch.bfloat16)
    y = ops.hgemm_tn(xa, sh.wt, b.to(DEV), relu=relu, out_bf16=out_bf16)
    want = (xa.double().cpu() @ _bf(w) + b.double())
    if relu:
        want = want.relu()
    assert y.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    assert rel_err(y.float(), want) < (5e-3 if out_bf16 else 3e-6)
    routed = bool(_lib.lib()._cdll.ttsmi_hgemm_k256_eligible(M, N, K))
    if M >= 4096:
        assert routed                                       # the decoder-size launches take the new kernel by default


@pytest.mark.parametrize('M', [700, 5000, 17001])
def test_weight_stationary_k256_gemm_mask_and_accumulate(M):
    """The ReLU'-masked bf16 dgrad (FFN2 -> hidden) and the accumulating fp32 dgrad (Wo top half) on gemm_k256.hip."""
    ops = _ops()
    K, N = 256, 1024
    dy = g(M, K, seed=1).to(DEV).to(torch.bfloat16)
    w2 = g(N, K, seed=2, scale=0.1)                                # FFN2 weight [F, d]: dgrad operand as stored
    sh = ops.make_shadow(w2.to(DEV))
    h1 = (g(M, N, seed=3)).to(DEV).to(torch.bfloat16)
    h1[::7, ::5] = 0                                               # exact zeros and negatives both mask
    h1[1::11, 3::7] = -0.0
    dh1 = ops.hgemm_tn(dy, sh.wb, relu_src=h1, out_bf16=True)
    want = (dy.double().cpu() @ _bf(w2).T) * (h1.double().cpu() > 0)
    assert rel_err(dh1.float(), want) < 5e-3
    # fp32 mask-free accumulate, N = 256
    wo = g(256, K, seed=4, scale=0.1)
    sho = ops.make_shadow(wo.to(DEV))
    acc0 = g(M, 256, seed=5).to(DEV)
    acc = acc0.clone()
    ops.hgemm_tn(dy, sho.wb, out=acc, accumulate=True)
    assert rel_err(acc, acc0.double().cpu() + dy.double().cpu() @ _bf(wo).T) < 3e-6
    # fp32 output with mask (not used by the model, covered for the ABI)
    out = ops.hgemm_tn(dy, sh.wb, relu_src=h1)
    assert rel_err(out, want) < 3e-6


@pytest.mark.parametrize('M', [700, 5000, 28800])
def test_k256_split_output_gemm(M):
    """ttsmi_hgemm_k256_split: one launch for d(h) += d_o.Wo_top^T (fp32 accumulate) and d(ctx) = d_o.Wo_ctx^T (bf16),
    against the two products computed in fp64 from the same bf16 operands; row tail (M % 64 != 0)."""
    ops = _ops()
    from transformertts_amd import _lib
    from transformertts_amd.ops import _p, _stream, check
    l = _lib.lib()
    d = 256
    d_o = g(M, d, seed=1).to(DEV).to(torch.bfloat16)
    wo = g(2 * d, d, seed=2, scale=0.1)                              # Wo as stored [2d, d]
    sh = ops.make_shadow(wo.to(DEV))
    dh0 = g(M, d, seed=3).to(DEV)
    dh = dh0.clone()
    dctx = torch.full((M, d), float('nan'), device=DEV, dtype=torch.bfloat16)
    check(l.ttsmi_hgemm_k256_split(_p(d_o), d, _p(sh.wb), d, _p(dh), d, d, _p(dctx), d, M, 2 * d, _stream()))
    torch.cuda.synchronize()
    a = d_o.double().cpu()
    assert rel_err(dh, dh0.double().cpu() + a @ _bf(wo[:d]).T) < 3e-6
    assert rel_err(dctx.float(), a @ _bf(wo[d:]).T) < 5e-3
    with pytest.raises(_lib.TtsmiError):
        check(l.ttsmi_hgemm_k256_split(_p(d_o), d, _p(sh.wb), d, _p(dh), d, 100, _p(dctx), d, M, 2 * d, _stream()))


@pytest.mark.parametrize('B,H,T,dh,pdrop', [(2, 4, 333, 64, 0.0), (2, 2, 200, 32, 0.2), (1, 2, 150, 192, 0.1), (3, 4, 900, 64, 0.1)])
def test_bf16_attention_maps_equal_the_fp32_recomputation(B, H, T, dh, pdrop):
    """ttsmi_attention_weights with TTSMI_BF16_IO (bf16 MFMA on the bf16 qkv the forward read) against the exact-fp32 kernel
    on the widened copy of the same tensor: same softmax (the forward's log-sum-exp), same keep decisions - dropped
    entries are exactly 0 in both - ragged padding, T not a multiple of the tile."""
    ops = _ops()
    from transformertts_amd import _lib
    d = H * dh
    qkv = (g(B * T, 3 * d, seed=1) * 0.6).to(DEV).to(torch.bfloat16)
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 1)) for i in range(B - 1)])
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    pad[0, 3] = 1
    klen = torch.tensor([int((p == 0).nonzero().max()) + 1 for p in pad], dtype=torch.int32)
    pad, klen = pad.to(DEV), klen.to(DEV)
    step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    drop = ops.DropCtx(seed=5, step_dev=step)
    _ctx, lse = ops.AttentionFn.apply(qkv.float(), pad, klen, B, H, T, dh, 0.0, None, 0, _lib.TTSMI_BF16)
    w16 = ops.attention_weights(qkv, pad, lse, B, H, T, dh, pdrop, drop, 7, _lib.TTSMI_BF16_IO)
    w32 = ops.attention_weights(qkv.float(), pad, lse, B, H, T, dh, pdrop, drop, 7)
    torch.cuda.synchronize()
    assert w16.shape == (B, H, T, T) and torch.isfinite(w16).all()
    assert torch.equal(w16 == 0, w32 == 0)                                  # identical keep decisions / padded keys
    assert rel_err(w16, w32) < 2e-5                                          # same products, different summation order
    if pdrop > 0:
        # the keep decisions read from the layer's bit table (what a training step with maps uses): the same numbers
        table = ops.attention_dropmask(B, H, T, pdrop, drop, 7, DEV)
        wbits = ops.attention_weights(qkv, pad, lse, B, H, T, dh, pdrop, drop, 7, _lib.TTSMI_BF16_IO, table)
        torch.cuda.synchronize()
        assert torch.equal(wbits, w16)
    if pdrop == 0.0:
        rows = w16.sum(-1)
        assert float((rows - 1).abs().max()) < 2e-2                          # lse comes from the bf16 forward


@pytest.mark.parametrize('B,H,T,dh', [(2, 2, 150, 192), (2, 4, 333, 64)])
def test_keep_bit_table_with_fp32_tensors_equals_the_hashed_dropout(B, H, T, dh):
    """AttentionFn on fp32 tensors with the bf16 MFMA kernels (TTSMI_BF16: the conv-block path of the reference-default
    architecture): the keep-bit table and the in-kernel hash make the same decisions - context, log-sum-exp and dqkv
    agree to rounding order."""
    ops = _ops()
    from transformertts_amd._lib import TTSMI_BF16
    d, pdrop = H * dh, 0.2
    qkv = (g(B * T, 3 * d, seed=1) * 0.6).to(torch.bfloat16).float()
    dctx = g(B * T, d, seed=2).to(torch.bfloat16).float().to(DEV)
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 1)) for i in range(B - 1)])
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    klen = torch.tensor([int((p == 0).nonzero().max()) + 1 for p in pad], dtype=torch.int32)
    pad, klen = pad.to(DEV), klen.to(DEV)
    step = torch.full((1,), 2, dtype=torch.int64, device=DEV)
    drop = ops.DropCtx(seed=9, step_dev=step)
    outs = []
    for masked in (False, True):
        q = qkv.to(DEV).requires_grad_()
        m = ops.attention_dropmask(B, H, T, pdrop, drop, 6, DEV) if masked else None
        ctx, lse = ops.AttentionFn.apply(q, pad, klen, B, H, T, dh, pdrop, drop, 6, TTSMI_BF16, m)
        ctx.backward(dctx)
        outs.append((ctx.detach().cpu(), lse.cpu(), q.grad.cpu()))
    (c0, l0, g0), (c1, l1, g1) = outs
    assert torch.equal(l0, l1)
    assert rel_err(c1, c0) < 1e-5 and rel_err(g1, g0) < 1e-2
    assert float((g1 - g0).abs().mean()) < 2e-3 * float(g0.abs().mean())
