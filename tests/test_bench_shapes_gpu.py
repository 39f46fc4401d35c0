"""-m gpu parity of the kernel VARIANTS THE BENCHMARK LAUNCHES, at the sizes it launches them, against fp64 references.

Routing inside libttsmi is size dependent (rowgemm.hip: the 128-row LDS-DMA full-row kernel from M >= 16 257 rows;
gemm_bf16.hip: the persistent LDS-DMA GEMM from 192 tiles of 128 x 128 and K >= 512; the LDS-DMA weight-gradient
kernel; the keep-bit-table attention kernels), so the small-shape tests of test_ops_gpu.py never reach the
instantiations that bench.py's configs[1] step (B 32: M_dec = 28 800, M_enc = 6 400) runs.  Every case here
  * runs at the benchmark's row count (and at 16 384 + a ragged tail, the first size that takes the big variant),
  * asserts through ttsmi_last_kernel() that the variant named in profiles/r0*_bench_bf16_kernel_stats.csv ran,
  * compares with an fp64 torch-CPU evaluation of the reference's formula (model/layers.py:82-102 FFNResNorm,
    :198-211 SelfAttentionResNorm, :148-150 output projection, :176-195 scaled-dot-product attention) on the same
    bf16-rounded operands, with dropout ON: the fp64 side applies the keep mask restated in tests/_dropout_ref.py.
What remains is the fp32 accumulation order and the bf16 rounding of stored outputs; bounds are stated per output."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _dropout_ref as dr  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
EPS = 1e-6


def _env():
    from transformertts_amd import _lib, ops
    return ops, _lib, _lib.lib()


def last_kernel(l) -> str:
    return l.ttsmi_last_kernel().decode()


def rel_err(got, want) -> float:
    want = want.double().cpu()
    got = got.double().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-30)


def mean_err(got, want) -> float:
    want = want.double().cpu()
    got = got.double().cpu()
    return float((got - want).abs().mean()) / max(float(want.abs().mean()), 1e-30)


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=gen) * scale).float()


def bf(x):
    return x.to(torch.bfloat16).double()


def test_dropout_restatement_matches_the_library_hash():
    """tests/_dropout_ref.py against the kernels: LayerNorm-with-input-dropout of a constant tensor (res = 0,
    gamma = 1, beta = 0) has x^ < 0 exactly where the element was dropped (dropped -> 0 < row mean, kept -> 1/(1-p))."""
    ops, _lib, l = _env()
    M, C, p, seed, stepv, site = 777, 256, 0.3, 1234567, 9, 11
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    drop = ops.DropCtx(seed=seed, step_dev=step)
    x = torch.ones(M, C, device=DEV)
    y, _yh, _mean, _rstd = ops._ln_fwd(x, torch.zeros_like(x), torch.ones(C, device=DEV), torch.zeros(C, device=DEV), None,
                                       p, site, drop, True)
    torch.cuda.synchronize()
    keep = dr.keep_mask(seed, stepv, site, np.arange(M), np.arange(C), p)
    assert 0.65 < keep.mean() < 0.75
    assert np.array_equal((y.cpu().numpy() > 0), keep)


def _ln_ref(z, gamma, beta):
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + EPS)
    xh = (z - mu) * rstd
    return xh * gamma + beta, xh, rstd[:, 0]


# (M, K, dual): the output projection Dense(concat([q_in, ctx])) = K 512 in two A segments, FFN2 = K 1024
@pytest.mark.parametrize('M,K,dual', [(28800, 512, True), (28800, 1024, False), (16384 + 77, 512, True), (16384 + 77, 1024, False)])
def test_fused_gemm_layernorm_forward_at_the_benchmark_rows(M, K, dual):
    """ttsmi_hgemm_ln_fwd = rowgemm_dma_kernel<0, 128>:  y = rowmask(LN(dropout(a.W + b) + res)) and its bf16 copy,
    x^ and rstd (reference model/layers.py:96-102, 207-211 with the row mask of :229-230)."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    N, pdrop, seed, stepv, site = 256, 0.1, 4242, 3, 5
    a = g(M, K, seed=1).to(torch.bfloat16)
    w = g(K, N, seed=2, scale=0.05)
    bias, gam, bet = g(N, seed=3), 1 + 0.1 * g(N, seed=4), 0.1 * g(N, seed=5)
    res = g(M, N, seed=6)
    pad = (torch.arange(M) % 7 == 3).to(torch.uint8)
    # fp64 reference
    keep = torch.from_numpy(dr.keep_mask(seed, stepv, site, np.arange(M), np.arange(N), pdrop))
    z = (a.double() @ bf(w) + bias.double()) * keep * (1.0 / (1.0 - float(np.float32(pdrop)))) + res.double()
    y_ref, xh_ref, rstd_ref = _ln_ref(z, gam.double(), bet.double())
    live = pad == 0
    y_ref = y_ref * live[:, None]
    # device
    ad, wd = a.to(DEV), w.to(DEV)
    sh = ops.make_shadow(wd)
    a1, a2 = (ad[:, :K // 2].contiguous(), ad[:, K // 2:].contiguous()) if dual else (ad, None)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    y = torch.empty(M, N, device=DEV)
    yh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    xh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    rstd = torch.empty(M, device=DEV)
    bias_d, res_d, gam_d, bet_d, pad_d = (t.to(DEV) for t in (bias, res, gam, bet, pad))     # (kept alive over the launch)
    check(l.ttsmi_hgemm_ln_fwd(_p(a1), a1.stride(0), _p(a2), 0 if a2 is None else a2.stride(0), a1.shape[1] if dual else 0,
                               _p(sh.wt), sh.wt.stride(0), _p(bias_d), _p(res_d), _p(gam_d), _p(bet_d),
                               _p(pad_d), pdrop, site, seed, _p(step), EPS, _p(y), _p(yh), _p(xh), _p(rstd), M, N, K,
                               _stream()))
    torch.cuda.synchronize()
    assert last_kernel(l) == 'rowgemm_dma_kernel<0, 128>'
    assert rel_err(y, y_ref) < 2e-5                                  # fp32 accumulation order only
    assert rel_err(rstd, rstd_ref) < 2e-5
    assert rel_err(yh.float(), y_ref) < 4e-3                          # + one bf16 rounding (2^-9 of the value)
    assert rel_err(xh.float()[live.to(DEV)], xh_ref[live]) < 4e-3
    assert float(y[(~live).to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize('M,kernel', [(28800, 'rowgemm_dma_kernel<0, 128, 1>'), (6400, 'rowgemm_dma_kernel<0, 64, 1>')])
@pytest.mark.parametrize('K,dual,want_y', [(512, True, False), (1024, False, True)])
def test_fused_gemm_layernorm_forward_with_a_bf16_residual(M, kernel, K, dual, want_y):
    """ttsmi_hgemm_ln_fwd_h, the form the planned dense blocks launch since the residual stream inside a stack is bf16
    (ttsmi_dense_block.res16): the residual is the bf16 tensor the GEMM reads as well (K 512: [h | ctx].Wo + h with the
    residual = the first A segment), the fp32 y is written only on request (K 1024: the last block of a stack)."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    N, pdrop, seed, stepv, site = 256, 0.1, 77, 2, 9
    a = g(M, K, seed=1).to(torch.bfloat16)
    w = g(K, N, seed=2, scale=0.05)
    bias, gam, bet = g(N, seed=3), 1 + 0.1 * g(N, seed=4), 0.1 * g(N, seed=5)
    res = a[:, :N].contiguous() if dual else g(M, N, seed=6).to(torch.bfloat16)
    pad = (torch.arange(M) % 7 == 3).to(torch.uint8)
    keep = torch.from_numpy(dr.keep_mask(seed, stepv, site, np.arange(M), np.arange(N), pdrop))
    z = (a.double() @ bf(w) + bias.double()) * keep * (1.0 / (1.0 - float(np.float32(pdrop)))) + res.double()
    y_ref, xh_ref, rstd_ref = _ln_ref(z, gam.double(), bet.double())
    live = pad == 0
    y_ref = y_ref * live[:, None]
    ad, wd = a.to(DEV), w.to(DEV)
    sh = ops.make_shadow(wd)
    a1, a2 = (ad[:, :K // 2].contiguous(), ad[:, K // 2:].contiguous()) if dual else (ad, None)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    y = torch.full((M, N), 7.0, device=DEV)
    yh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    xh = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    rstd = torch.empty(M, device=DEV)
    bias_d, gam_d, bet_d, pad_d = (t.to(DEV) for t in (bias, gam, bet, pad))
    res_d = a1 if dual else res.to(DEV)
    check(l.ttsmi_hgemm_ln_fwd_h(_p(a1), a1.stride(0), _p(a2), 0 if a2 is None else a2.stride(0), a1.shape[1] if dual else 0,
                                 _p(sh.wt), sh.wt.stride(0), _p(bias_d), _p(res_d), _p(gam_d), _p(bet_d), _p(pad_d), pdrop, site,
                                 seed, _p(step), EPS, _p(y) if want_y else None, _p(yh), _p(xh), _p(rstd), M, N, K, _stream()))
    torch.cuda.synchronize()
    assert last_kernel(l) == kernel
    assert rel_err(rstd, rstd_ref) < 2e-5
    assert rel_err(yh.float(), y_ref) < 4e-3
    assert rel_err(xh.float()[live.to(DEV)], xh_ref[live]) < 4e-3
    if want_y:
        assert rel_err(y, y_ref) < 2e-5
        assert float(y[(~live).to(DEV)].abs().max()) == 0.0
    else:
        assert float((y - 7.0).abs().max()) == 0.0                    # nothing written through the NULL fp32 output


@pytest.mark.parametrize('M,tag', [(28800, '128'), (6400, '64')])
@pytest.mark.parametrize('dres16', [True, False])
def test_fused_dgrad_layernorm_backward_with_bf16_residual_gradients(M, tag, dres16):
    """ttsmi_hgemm_ln_bwd_dual_h: the upstream partial gradient arrives as bf16 and the residual gradient leaves as bf16
    (chained blocks) or fp32 (the bottom block of a stack); two K segments as the chained launch has them."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    N, K1, K2, pdrop, seed, stepv, site = 256, 768, 256, 0.1, 5, 9, 3
    a1 = g(M, K1, seed=1, scale=0.3).to(torch.bfloat16)
    a2 = g(M, K2, seed=2, scale=0.3).to(torch.bfloat16)
    w1, w2 = g(N, K1, seed=3, scale=0.05), g(2 * N, K2, seed=4, scale=0.05)
    part = g(M, N, seed=10).to(torch.bfloat16)
    xh = g(M, N, seed=11).to(torch.bfloat16)
    rstd = (0.5 + torch.rand(M, generator=torch.Generator().manual_seed(12))).float()
    gam = 1 + 0.1 * g(N, seed=4)
    pad = (torch.arange(M) % 5 == 1).to(torch.uint8)
    live = (pad == 0).double()[:, None]
    keep = torch.from_numpy(dr.keep_mask(seed, stepv, site, np.arange(M), np.arange(N), pdrop))
    dy = (part.double() + a1.double() @ bf(w1).T + a2.double() @ bf(w2)[:N].T) * live
    t = dy * gam.double()
    x = xh.double()
    dz = rstd.double()[:, None] * (t - t.mean(-1, keepdim=True) - x * (t * x).mean(-1, keepdim=True))
    dx_ref = dz * keep * (1.0 / (1.0 - float(np.float32(pdrop))))
    dg_ref, db_ref = (dy * x).sum(0), dy.sum(0)
    w1d, w2d = w1.to(DEV).to(torch.bfloat16), w2.to(DEV).to(torch.bfloat16)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    dxb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    dres = torch.empty(M, N, device=DEV, dtype=torch.bfloat16 if dres16 else torch.float32)
    nw = int(l.ttsmi_hgemm_ln_bwd_nparts(M))
    ws = torch.empty(int(l.ttsmi_layernorm_partials_bytes(nw, N)), dtype=torch.uint8, device=DEV)
    a1d, a2d, part_d, xh_d, rstd_d, gam_d, pad_d = (t.to(DEV) for t in (a1, a2, part, xh, rstd, gam, pad))
    check(l.ttsmi_hgemm_ln_bwd_dual_h(_p(a1d), K1, _p(a2d), K2, K1, _p(w1d), K1, _p(w2d), K2, _p(part_d), _p(xh_d), _p(rstd_d),
                                      _p(gam_d), _p(pad_d), pdrop, site, seed, _p(step), _p(dxb), _p(dres), int(dres16), _p(ws),
                                      ws.numel(), M, N, K1 + K2, _stream()))
    assert last_kernel(l) == f'rowgemm_dma_kernel<1, {tag}, {3 if dres16 else 1}>'
    dg, db = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    with ops.ln_param_batch():
        ops._ln_defer(ws, dg, db, None, M, N, nw)
    torch.cuda.synchronize()
    assert rel_err(dres.float(), dz) < (4e-3 if dres16 else 2e-5)
    assert rel_err(dxb.float(), dx_ref) < 4e-3
    assert rel_err(dg, dg_ref) < 1e-4 and rel_err(db, db_ref) < 1e-4
    # argument errors stay errors in the bf16 forms: a null upstream gradient
    assert l.ttsmi_hgemm_ln_bwd_dual_h(_p(a1d), K1, _p(a2d), K2, K1, _p(w1d), K1, _p(w2d), K2, None, _p(xh_d), _p(rstd_d),
                                       _p(gam_d), _p(pad_d), pdrop, site, seed, _p(step), _p(dxb), _p(dres), int(dres16), _p(ws),
                                       ws.numel(), M, N, K1 + K2, _stream()) != 0


def test_layernorm_backward_xhat_with_a_bf16_residual_gradient():
    """ttsmi_layernorm_bwd_xhat_h (the top block of a res16 stack): the same backward with dres stored as bf16."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    M, N, pdrop, seed, stepv, site = 28800, 256, 0.1, 5, 9, 3
    dy = g(M, N, seed=10)
    xh = g(M, N, seed=11).to(torch.bfloat16)
    rstd = (0.5 + torch.rand(M, generator=torch.Generator().manual_seed(12))).float()
    gam = 1 + 0.1 * g(N, seed=4)
    pad = (torch.arange(M) % 5 == 1).to(torch.uint8)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    nw = int(l.ttsmi_layernorm_bwd_xhat_nparts(M))
    ws = torch.empty(int(l.ttsmi_layernorm_partials_bytes(nw, N)), dtype=torch.uint8, device=DEV)
    dy_d, xh_d, rstd_d, gam_d, pad_d = (t.to(DEV) for t in (dy, xh, rstd, gam, pad))
    outs = {}
    for h in (False, True):
        dxb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        dres = torch.empty(M, N, device=DEV, dtype=torch.bfloat16 if h else torch.float32)
        fn = l.ttsmi_layernorm_bwd_xhat_h if h else l.ttsmi_layernorm_bwd_xhat
        check(fn(_p(dy_d), _p(xh_d), _p(rstd_d), _p(gam_d), _p(pad_d), pdrop, site, seed, _p(step), _p(dxb), _p(dres), _p(ws),
                 ws.numel(), M, N, _stream()))
        torch.cuda.synchronize()
        outs[h] = (dxb, dres)
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][1], outs[False][1].to(torch.bfloat16))      # the fp32 result, rounded once


# K 1024 = FFN1 dgrad + res-norm 1; K 768 = the qkv dgrad of the block above chained into res-norm 2
@pytest.mark.parametrize('M,K', [(28800, 1024), (28800, 768), (16384 + 77, 1024)])
def test_fused_dgrad_layernorm_backward_at_the_benchmark_rows(M, K):
    """ttsmi_hgemm_ln_bwd = rowgemm_dma_kernel<1, 128>: dy = dy_part + a.W^T, then the x^-form LayerNorm backward
    dz = rstd (t - mean t - x^ mean(t x^)), t = dy.rowmask.gamma; outputs dres = dz (fp32), dx = dropout'(dz) (bf16)
    and the per-workgroup dgamma / dbeta partials (backward of model/layers.py:96-102 / :207-211)."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    N, pdrop, seed, stepv, site = 256, 0.1, 99, 4, 7
    ab = g(M, K, seed=8, scale=0.3).to(torch.bfloat16)
    wb = g(N, K, seed=9, scale=0.05)                      # W as stored [256][K]: the dgrad operand
    part = g(M, N, seed=10)
    xh = g(M, N, seed=11).to(torch.bfloat16)
    rstd = (0.5 + torch.rand(M, generator=torch.Generator().manual_seed(12))).float()
    gam = 1 + 0.1 * g(N, seed=4)
    pad = (torch.arange(M) % 7 == 3).to(torch.uint8)
    live = (pad == 0).double()[:, None]
    keep = torch.from_numpy(dr.keep_mask(seed, stepv, site, np.arange(M), np.arange(N), pdrop))
    # fp64 reference
    dy = (part.double() + ab.double() @ bf(wb).T) * live
    t = dy * gam.double()
    x = xh.double()
    dz = rstd.double()[:, None] * (t - t.mean(-1, keepdim=True) - x * (t * x).mean(-1, keepdim=True))
    dx_ref = dz * keep * (1.0 / (1.0 - float(np.float32(pdrop))))
    dg_ref, db_ref = (dy * x).sum(0), dy.sum(0)
    # device
    wd = wb.to(DEV)
    sh = ops.make_shadow(wd)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    dxb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    dres = torch.empty(M, N, device=DEV)
    nw = int(l.ttsmi_hgemm_ln_bwd_nparts(M))
    ws = torch.empty(int(l.ttsmi_layernorm_partials_bytes(nw, N)), dtype=torch.uint8, device=DEV)
    abd, part_d, xh_d, rstd_d, gam_d, pad_d = (t.to(DEV) for t in (ab, part, xh, rstd, gam, pad))   # (kept alive over the launch)
    check(l.ttsmi_hgemm_ln_bwd(_p(abd), abd.stride(0), _p(sh.wb), sh.wb.stride(0), _p(part_d), _p(xh_d),
                               _p(rstd_d), _p(gam_d), _p(pad_d), pdrop, site, seed, _p(step), _p(dxb),
                               _p(dres), _p(ws), ws.numel(), M, N, K, _stream()))
    assert last_kernel(l) == 'rowgemm_dma_kernel<1, 128>'
    dg, db = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    with ops.ln_param_batch():
        ops._ln_defer(ws, dg, db, None, M, N, nw)
    torch.cuda.synchronize()
    assert rel_err(dres, dz) < 2e-5
    assert rel_err(dxb.float(), dx_ref) < 4e-3
    assert rel_err(dg, dg_ref) < 1e-4 and rel_err(db, db_ref) < 1e-4      # fp32 column sums over 28 800 rows


@pytest.mark.parametrize('M,kernel', [(28800, 'rowgemm_dma_kernel<1, 128>'), (6400, 'rowgemm_dma_kernel<1, 64>'),
                                      (1000, 'rowgemm_kernel<1>')])
def test_fused_dgrad_layernorm_backward_with_two_k_segments(M, kernel):
    """ttsmi_hgemm_ln_bwd_dual as the chained block backward launches it: dy = dy_part + dqkv.Wqkv^T + d_o.Wo[:d]^T with
    each K segment reading its own activation AND its own weight matrix (K = 768 + 256), then res-norm 2's backward of
    the block below (model/layers.py:116-118,148-149 backward into :100-102)."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    N, K1, K2, pdrop, seed, stepv, site = 256, 768, 256, 0.1, 5, 9, 3
    a1 = g(M, K1, seed=1, scale=0.3).to(torch.bfloat16)
    a2 = g(M, K2, seed=2, scale=0.3).to(torch.bfloat16)
    w1 = g(N, K1, seed=3, scale=0.05)                     # Wqkv as stored [256][768]
    w2 = g(2 * N, K2, seed=4, scale=0.05)                 # Wo as stored [512][256]: only its first 256 rows take part
    part = g(M, N, seed=10)
    xh = g(M, N, seed=11).to(torch.bfloat16)
    rstd = (0.5 + torch.rand(M, generator=torch.Generator().manual_seed(12))).float()
    gam = 1 + 0.1 * g(N, seed=4)
    pad = (torch.arange(M) % 5 == 1).to(torch.uint8)
    live = (pad == 0).double()[:, None]
    keep = torch.from_numpy(dr.keep_mask(seed, stepv, site, np.arange(M), np.arange(N), pdrop))
    dy = (part.double() + a1.double() @ bf(w1).T + a2.double() @ bf(w2)[:N].T) * live
    t = dy * gam.double()
    x = xh.double()
    dz = rstd.double()[:, None] * (t - t.mean(-1, keepdim=True) - x * (t * x).mean(-1, keepdim=True))
    dx_ref = dz * keep * (1.0 / (1.0 - float(np.float32(pdrop))))
    dg_ref, db_ref = (dy * x).sum(0), dy.sum(0)
    w1d, w2d = w1.to(DEV).to(torch.bfloat16), w2.to(DEV).to(torch.bfloat16)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    dxb = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    dres = torch.empty(M, N, device=DEV)
    nw = int(l.ttsmi_hgemm_ln_bwd_nparts(M))
    ws = torch.empty(int(l.ttsmi_layernorm_partials_bytes(nw, N)), dtype=torch.uint8, device=DEV)
    a1d, a2d, part_d, xh_d, rstd_d, gam_d, pad_d = (t.to(DEV) for t in (a1, a2, part, xh, rstd, gam, pad))
    check(l.ttsmi_hgemm_ln_bwd_dual(_p(a1d), K1, _p(a2d), K2, K1, _p(w1d), K1, _p(w2d), K2, _p(part_d), _p(xh_d),
                                    _p(rstd_d), _p(gam_d), _p(pad_d), pdrop, site, seed, _p(step), _p(dxb), _p(dres),
                                    _p(ws), ws.numel(), M, N, K1 + K2, _stream()))
    assert last_kernel(l) == kernel
    dg, db = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    with ops.ln_param_batch():
        ops._ln_defer(ws, dg, db, None, M, N, nw)
    torch.cuda.synchronize()
    assert rel_err(dres, dz) < 2e-5
    assert rel_err(dxb.float(), dx_ref) < 4e-3
    assert rel_err(dg, dg_ref) < 1e-4 and rel_err(db, db_ref) < 1e-4
    # one of the two operands of the second segment alone is an argument error, not a silent single-segment run
    assert l.ttsmi_hgemm_ln_bwd_dual(_p(a1d), K1, _p(a2d), K2, K1, _p(w1d), K1, None, 0, _p(part_d), _p(xh_d), _p(rstd_d),
                                     _p(gam_d), _p(pad_d), pdrop, site, seed, _p(step), _p(dxb), _p(dres), _p(ws),
                                     ws.numel(), M, N, K1 + K2, _stream()) != 0


@pytest.mark.parametrize('K', [512, 768, 1024])
@pytest.mark.parametrize('mode', ['plain', 'dual', 'accumulate', 'bf16out'])
def test_persistent_dma_gemm_at_the_benchmark_rows(K, mode):
    """ttsmi_hgemm_tn at (28 800, K, 256) = gemm_bf16_dma_kernel (decoder-size launches with K >= 512: the output
    projection, FFN2 and their dgrads), every epilogue the model uses, against the fp64 product of the same bf16
    operands (Dense forward / dgrad: model/layers.py:93-94,148-149)."""
    ops, _lib, l = _env()
    M, N = 28800, 256
    if mode == 'dual' and K != 512:
        pytest.skip('dual-A is the K = 512 output projection')
    a = g(M, K, seed=1).to(torch.bfloat16)
    w = g(K, N, seed=2, scale=0.05)
    b = g(N, seed=3)
    sh = ops.make_shadow(w.to(DEV))
    ad = a.to(DEV)
    want = a.double() @ bf(w)
    if mode == 'plain':
        y = ops.hgemm_tn(ad, sh.wt, b.to(DEV), relu=True)
        want = (want + b.double()).relu()
    elif mode == 'dual':
        y = ops.hgemm_tn(ad[:, :256].contiguous(), sh.wt, b.to(DEV), False, ad[:, 256:].contiguous())
        want = want + b.double()
    elif mode == 'accumulate':
        acc0 = g(M, N, seed=5)
        y = acc0.to(DEV)
        ops.hgemm_tn(ad, sh.wt, out=y, accumulate=True)
        want = want + acc0.double()
    else:
        y = ops.hgemm_tn(ad, sh.wt, b.to(DEV), out_bf16=True)
        want = want + b.double()
    torch.cuda.synchronize()
    assert last_kernel(l) == 'gemm_bf16_dma_kernel'
    assert rel_err(y.float(), want) < (4e-3 if mode == 'bf16out' else 3e-6)


@pytest.mark.parametrize('mode', ['conv-fwd', 'conv-dgrad', 'plain-f32', 'ragged'])
def test_256_tile_dma_gemm_at_the_reference_default_conv_shapes(mode):
    """ttsmi_hgemm_tn on the LARGE launches of the reference-default conv stacks = gemm_bf16_dma256_kernel (256 x 256
    tiles): (28 864, K 1 152, N 1 536) as the first conv's forward (overlapping A rows: lda = C, K = 3 C; bias, ReLU, bf16
    out) and the second conv's dgrad (bf16 ReLU' mask, bf16 out), a plain fp32-out product, and a shape whose M and N are
    not multiples of the tile; against the fp64 product of the same bf16 operands on a sample of rows
    (model/layers.py:19-26, 30-38 through ops.ConvStackFn's plain-GEMM route)."""
    ops, _lib, l = _env()
    C, k = 384, 3
    K, N = k * C, 1536
    M = 32 * 902 - 2 if mode != 'ragged' else 131 * 256 + 77
    if mode == 'ragged':
        N = 1536 - 64
    rows_buf = M + 2
    w = g(K, N, seed=2, scale=0.03)
    b = g(N, seed=3)
    sh = ops.make_shadow(w.to(DEV))
    idx = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M)])
    if mode in ('conv-fwd', 'conv-dgrad'):
        buf = g(rows_buf, C, seed=1).to(torch.bfloat16)
        ad = buf.to(DEV)
        a_view = torch.as_strided(ad, (M, K), (C, 1))                      # row j = buffer rows j .. j + 2
        a_ref = torch.as_strided(buf, (M, K), (C, 1))[idx].double()
    else:
        a = g(M, K, seed=1).to(torch.bfloat16)
        a_view = a.to(DEV)
        a_ref = a[idx].double()
    want = a_ref @ bf(w)
    if mode == 'conv-fwd':
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        ops.hgemm_tn(a_view, sh.wt, b.to(DEV), relu=True, out=out)
        want, tol = (want + b.double()).relu(), 4e-3
    elif mode == 'conv-dgrad':
        mask = g(M, N, seed=7).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
        ops.hgemm_tn(a_view, sh.wt, None, relu_src=mask.to(DEV), out=out)
        want, tol = want * (mask[idx].double() > 0), 4e-3
    else:
        out = ops.hgemm_tn(a_view, sh.wt, b.to(DEV))
        want, tol = want + b.double(), 3e-6
    torch.cuda.synchronize()
    assert last_kernel(l) == 'gemm_bf16_dma256_kernel'
    assert rel_err(out[idx.to(DEV)].float(), want) < tol
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize('C,N,B,T', [(384, 1536, 32, 900), (1536, 384, 32, 900), (128, 128, 4, 30)])
def test_conv_weight_gradient_with_shifted_row_taps(C, N, B, T):
    """ttsmi_hgemm_wgrad_rows with conv_taps = 3, conv_T = 0: the three taps of a 'same' Conv1D over the zero-margin layout
    [B (T + 2) + 2, C] as ONE wgrad_dma_kernel launch (rows [j C, (j + 1) C) of dW from x shifted down by j rows) - against
    the fp64 products on a sample of elements and against the three per-tap launches it replaces
    (model/layers.py:19-26 differentiated; ops.ConvStackFn._backward_plain)."""
    ops, _lib, l = _env()
    k = 3
    rows = B * (T + 2)
    x = torch.zeros(rows + 2, C)
    gy = torch.zeros(rows + 2, N)
    xv, gv = x[:rows].view(B, T + 2, C), gy[:rows].view(B, T + 2, N)
    xv[:, 1:T + 1] = g(B, T, C, seed=1)
    gv[:, 1:T + 1] = g(B, T, N, seed=2) * 0.1
    xh, gh = x.to(torch.bfloat16), gy.to(torch.bfloat16)
    xd, gd = xh.to(DEV), gh.to(DEV)
    dw = torch.full((k, C, N), float('nan'), device=DEV)
    db = torch.full((N,), float('nan'), device=DEV)
    ops.hgemm_wgrad_rows(xd[:rows], gd[1:1 + rows], dw.reshape(k * C, N), db, conv=(k, 0, C, 0))
    torch.cuda.synchronize()
    assert last_kernel(l) == 'wgrad_dma_kernel'
    dw2 = torch.full((k, C, N), float('nan'), device=DEV)
    db2 = torch.full((N,), float('nan'), device=DEV)
    for tap in range(k):
        ops.hgemm_wgrad_rows(xd[tap:tap + rows], gd[1:1 + rows], dw2[tap], db2 if tap == 0 else None)
    torch.cuda.synchronize()
    assert rel_err(dw, dw2) < 2e-6 and rel_err(db, db2) < 2e-6                  # the same products, other split sizes
    ci, ni = torch.arange(0, C, max(1, C // 37)), torch.arange(0, N, max(1, N // 41))
    gd64 = gh[1:1 + rows].double()
    for tap in range(k):
        want = xh[tap:tap + rows][:, ci].double().T @ gd64[:, ni]
        assert rel_err(dw[tap][ci][:, ni], want) < 3e-6, tap
    assert rel_err(db, gd64.sum(0)) < 3e-6
    with pytest.raises(Exception):                                            # fp32 x is the windowed form's business
        ops.hgemm_wgrad_rows(xd[:rows].float(), gd[1:1 + rows], dw.reshape(k * C, N), db, conv=(k, 0, C, 0))


@pytest.mark.parametrize('M,fwd_kernel,bwd_kernel', [(28800, 'gemm_k256_wide_kernel<1>', 'gemm_k256_wide_kernel<4>'),
                                                      (6400, 'gemm_k256_kernel<1>', 'gemm_k256_kernel<4>'),
                                                      (4096 + 37, 'gemm_k256_kernel<1>', 'gemm_k256_kernel<4>')])
def test_relu_bit_matrix_of_the_ffn(M, fwd_kernel, bwd_kernel):
    """ttsmi_hgemm_k256_relu_bits / _masked_bits (the FFN's ReLU handed to the backward as one bit per element,
    model/layers.py:99): h1 and the masked FFN2 dgrad must be BIT-IDENTICAL to the launches that store / re-read the bf16
    activation (ttsmi_hgemm_tn RELU / MASK_BF16, themselves checked against fp64 in test_ops_gpu)."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    d, F = 256, 1024
    a = g(M, d, seed=1).to(torch.bfloat16).to(DEV)
    w1 = ops.make_shadow(g(d, F, seed=2, scale=0.05).to(DEV))            # .wt = W1^T [F][d]
    b1 = g(F, seed=3).to(DEV)
    df = g(M, d, seed=4).to(torch.bfloat16).to(DEV)
    w2 = ops.make_shadow(g(F, d, seed=5, scale=0.05).to(DEV))            # .wb = W2 as stored [F][d]: the dgrad operand
    h1_ref = ops.hgemm_tn(a, w1.wt, b1, relu=True, out_bf16=True)
    dh1_ref = ops.hgemm_tn(df, w2.wb, None, relu_src=h1_ref, out_bf16=True)
    nbytes = int(l.ttsmi_relu_bits_bytes(M, F))
    assert nbytes >= M * F // 8
    bits = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    h1 = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    check(l.ttsmi_hgemm_k256_relu_bits(_p(a), d, _p(w1.wt), d, _p(b1), _p(h1), F, _p(bits), M, F, _stream()))
    assert last_kernel(l) == fwd_kernel
    dh1 = torch.empty(M, F, dtype=torch.bfloat16, device=DEV)
    check(l.ttsmi_hgemm_k256_masked_bits(_p(df), d, _p(w2.wb), d, _p(bits), _p(dh1), F, M, F, _stream()))
    assert last_kernel(l) == bwd_kernel
    torch.cuda.synchronize()
    assert torch.equal(h1.view(torch.int16), h1_ref.view(torch.int16))
    # (the bit order is the kernels' own: what is checked is that the backward reads exactly the mask the forward meant)
    assert 0.2 < float((h1_ref.float() > 0).float().mean()) < 0.8          # (a mask that is all ones or all zeros proves nothing)
    assert torch.equal(dh1.view(torch.int16), dh1_ref.view(torch.int16))
    # a shape the K = 256 kernel does not take is refused, not routed elsewhere
    assert l.ttsmi_hgemm_k256_relu_bits(_p(a), d, _p(w1.wt), d, _p(b1), _p(h1), F, _p(bits), 64, F, _stream()) == -3     # TTSMI_ERR_UNSUPPORTED


# the dense block's weight gradients (dense_block.hip): FFN2 [1024 -> 256], FFN1 [256 -> 1024], the two halves of Wo
# [256 -> 256] (the second without a bias gradient), Wqkv [256 -> 768]; all operands bf16.  Last case: an fp32 x takes
# the register-staged kernel (the route assertion must be able to fail).
@pytest.mark.parametrize('K,N,xh,dyh,bias', [(1024, 256, True, True, True), (256, 1024, True, True, True),
                                             (256, 256, True, True, True), (256, 256, True, True, False),
                                             (256, 768, True, True, True), (256, 1024, False, True, True)])
@pytest.mark.parametrize('M', [28800, 9100, 12345, 6400 + 31, 33, 7])
def test_weight_gradient_dma_kernel_at_the_benchmark_rows(K, N, xh, dyh, bias, M):
    """ttsmi_hgemm_wgrad_rows at M = 28 800 = wgrad_dma_kernel (+ the slab reduction and the bias gradient) against
    fp64 on the same bf16 operands (tape.gradient of the Dense layers, model/models.py:480); row counts that are not
    multiples of the kernel's 32-row step (the reference's bucketed batches: 14 x 650, ...) take the same kernel - its last
    step zeroes the fragments of the missing rows - behind operands whose rows past M are NaN."""
    ops, _lib, l = _env()
    if M != 28800 and (K, N, bias) not in ((1024, 256, True), (256, 768, True), (256, 256, False), (256, 1024, True)):
        pytest.skip('ragged row counts: one case per weight shape')
    x, dy = g(M, K, seed=1), g(M, N, seed=2, scale=0.2)
    # (views of larger buffers whose rows past M are NaN: a kernel that touched them would poison the result)
    xbuf = torch.full((M + 40, K), float('nan'), device=DEV, dtype=torch.bfloat16 if xh else torch.float32)
    dybuf = torch.full((M + 40, N), float('nan'), device=DEV, dtype=torch.bfloat16 if dyh else torch.float32)
    xd, dyd = xbuf[:M], dybuf[:M]
    xd.copy_(x.to(DEV))
    dyd.copy_(dy.to(DEV))
    dw, db = torch.empty(K, N, device=DEV), (torch.empty(N, device=DEV) if bias else None)
    ops.hgemm_wgrad_rows(xd, dyd, dw, db)
    torch.cuda.synchronize()
    assert last_kernel(l) == ('wgrad_dma_kernel' if (xh and dyh) else 'wgrad_rows_kernel')
    assert rel_err(dw, bf(x).T @ bf(dy)) < 5e-6
    if bias:
        assert rel_err(db, bf(dy).sum(0)) < 5e-6


def _attention_ref_chunk(qkv, pad, keep, H, T, dh, inv_keep, dctx):
    """fp64 scaled-dot-product attention with the additive -1e9 mask and inverted dropout on the weights
    (model/layers.py:176-195) for a chunk of samples; returns ctx and d(qkv) for upstream dctx."""
    Bc = qkv.shape[0] // T
    d = H * dh
    qd = qkv.double().requires_grad_()
    q, k, v = [t.reshape(Bc, T, H, dh).permute(0, 2, 1, 3) for t in qd.split(d, dim=1)]
    logits = q @ k.transpose(-1, -2) / (dh ** 0.5) + pad.double()[:, None, None, :] * -1e9
    w = torch.softmax(logits, -1) * keep * inv_keep
    ctx = (w @ v).permute(0, 2, 1, 3).reshape(Bc * T, d)
    ctx.backward(dctx.double())
    return ctx.detach(), qd.grad


def test_keep_bit_attention_at_the_benchmark_shape():
    """hattn_fwd / hattn_bwd_dq / hattn_bwd_dkv <64, 2, true> (bf16 I/O, keep-bit table) at (B, H, T, dh) =
    (32, 4, 900, 64) with dropout 0.1 and ragged key padding, forward context and d(qkv) against fp64 with the SAME keep
    decisions (restated hash), sample chunk by sample chunk."""
    ops, _lib, l = _env()
    from transformertts_amd.ops import _p, _stream, check
    B, H, T, dh, pdrop, seed, stepv, site = 32, 4, 900, 64, 0.1, 777, 6, 4
    d = H * dh
    qkv = (g(B * T, 3 * d, seed=1) * 0.7).to(torch.bfloat16)
    dctx = (g(B * T, d, seed=2) * 0.3).to(torch.bfloat16)
    lens = torch.tensor([T] + [max(1, (T * (i + 1)) // (B + 1)) for i in range(B - 1)])
    pad = (torch.arange(T)[None, :] >= lens[:, None]).to(torch.uint8)
    pad[0, 3] = 1
    klen = torch.tensor([int((p == 0).nonzero().max()) + 1 for p in pad], dtype=torch.int32)
    step = torch.full((1,), stepv, dtype=torch.int64, device=DEV)
    drop = ops.DropCtx(seed=seed, step_dev=step)
    qd, dd, padd, klend = qkv.to(DEV), dctx.to(DEV), pad.to(DEV), klen.to(DEV)
    m = ops.attention_dropmask(B, H, T, pdrop, drop, site, DEV)
    ctx = torch.empty(B * T, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, T, device=DEV)
    dqkv = torch.empty_like(qd)
    ws = torch.empty(int(l.ttsmi_attention_bwd_ws_bytes(B, H, T, dh)), dtype=torch.uint8, device=DEV)
    check(l.ttsmi_attention_fwd_masked(_p(qd), _p(padd), _p(klend), _p(ctx), _p(lse), B, H, T, dh, pdrop, _p(m),
                                       _lib.TTSMI_BF16_IO, _stream()))
    assert last_kernel(l) == 'hattn_fwd_kernel<64, 2, true>'
    check(l.ttsmi_attention_bwd_masked(_p(qd), _p(padd), _p(klend), _p(ctx), _p(dd), _p(lse), _p(dqkv), B, H, T, dh, pdrop,
                                       _p(m), _p(ws), ws.numel(), _lib.TTSMI_BF16_IO, _stream()))
    assert last_kernel(l) == 'hattn_bwd_dkv_kernel<64, 2, true>'
    torch.cuda.synchronize()
    # bit-reproducible: a second forward + backward on the same inputs gives the same bits (fixed summation orders, no atomics)
    ctx2, lse2, dq2 = torch.empty_like(ctx), torch.empty_like(lse), torch.full_like(qd, float('nan'))
    check(l.ttsmi_attention_fwd_masked(_p(qd), _p(padd), _p(klend), _p(ctx2), _p(lse2), B, H, T, dh, pdrop, _p(m),
                                       _lib.TTSMI_BF16_IO, _stream()))
    check(l.ttsmi_attention_bwd_masked(_p(qd), _p(padd), _p(klend), _p(ctx2), _p(dd), _p(lse2), _p(dq2), B, H, T, dh, pdrop,
                                       _p(m), _p(ws), ws.numel(), _lib.TTSMI_BF16_IO, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(ctx2.view(torch.int16), ctx.view(torch.int16)) and torch.equal(lse2, lse)
    assert torch.equal(dq2.view(torch.int16), dqkv.view(torch.int16))
    ctx, dqkv = ctx.float().cpu(), dqkv.float().cpu()
    inv_keep = 1.0 / (1.0 - float(np.float32(pdrop)))
    worst_c = worst_g = 0.0
    mean_c = mean_g = 0.0
    CH = 4
    for b0 in range(0, B, CH):
        rows = np.arange(b0 * H * T, (b0 + CH) * H * T)
        keep = torch.from_numpy(dr.keep_mask(seed, stepv, site, rows, np.arange(T), pdrop)).reshape(CH, H, T, T)
        sl = slice(b0 * T, (b0 + CH) * T)
        c_ref, g_ref = _attention_ref_chunk(qkv[sl], pad[b0:b0 + CH], keep, H, T, dh, inv_keep, dctx[sl])
        worst_c = max(worst_c, rel_err(ctx[sl], c_ref))
        worst_g = max(worst_g, rel_err(dqkv[sl], g_ref))
        mean_c = max(mean_c, mean_err(ctx[sl], c_ref))
        mean_g = max(mean_g, mean_err(dqkv[sl], g_ref))
    # bf16 rounding of P / dS / stored outputs: ~2^-9 per element, max over 7.4 M (22 M) elements
    assert worst_c < 1e-2 and worst_g < 2e-2, (worst_c, worst_g)
    # a wrong keep decision or a mis-indexed tile moves elements by O(1) of their value: the MEAN error stays at rounding
    assert mean_c < 4e-3 and mean_g < 6e-3, (mean_c, mean_g)
