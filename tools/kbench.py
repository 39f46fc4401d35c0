#!/usr/bin/env python
"""Kernel micro-benchmarks at the launch shapes of the configs[1] train step (decoder M = 28 800 rows, encoder
M = 6 400), one table per environment-knob variant so that A/B runs cost ONE gpurun call:

    python tools/kbench.py                       # default knobs
    python tools/kbench.py --variants base TTSMI_HGEMM_OCC4=1 TTSMI_HGEMM_DMA=1 ...
    python tools/kbench.py --only gemm|attn|ln   # a subset

Each variant runs in its own subprocess (the library reads its knobs once).  Every launch rotates over
several operand sets (working set > the 256 MB Infinity Cache would be unrealistic for the step, whose
producer just wrote the operand: 3 sets ~ what the step sees).  Times are HIP-event averages in us; TF/s
and TB/s use algorithmic FLOPs / bytes."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, n=30, warm=4):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def bench_gemm(rows_list=(28800, 6400)):
    import torch
    from transformertts_amd import ops
    dev, out = 'cuda:0', []
    bf, f32 = torch.bfloat16, torch.float32
    # name, K, N, A dtype, A2 (dual-A second segment width), out dtype, relu, relu_src dtype, accumulate
    cases = [('qkv  fwd', 256, 768, bf, 0, bf, False, None, False),
             ('o    fwd', 512, 256, bf, 256, f32, False, None, False),
             ('ffn1 fwd', 256, 1024, bf, 0, bf, True, None, False),
             ('ffn2 fwd', 1024, 256, bf, 0, f32, False, None, False),
             ('dh1  bwd', 256, 1024, bf, 0, bf, False, bf, False),
             ('da   bwd', 1024, 256, bf, 0, f32, False, None, True),
             ('dhto bwd', 256, 256, bf, 0, f32, False, None, True),
             ('dctx bwd', 256, 256, bf, 0, bf, False, None, False),
             ('dqkv bwd', 768, 256, bf, 0, f32, False, None, True)]
    R = 3
    for M in rows_list:
        for name, K, N, adt, k2, odt, relu, mdt, acc in cases:
            b = (torch.randn(N, K, device=dev) * 0.05).to(bf)
            bias = torch.randn(N, device=dev)
            k1 = K - k2
            As = [torch.randn(M, k1, device=dev).to(adt) for _ in range(R)]
            A2s = [torch.randn(M, k2, device=dev).to(adt) for _ in range(R)] if k2 else None
            Os = [torch.zeros(M, N, device=dev, dtype=odt) for _ in range(R)]
            Ms = [torch.randn(M, N, device=dev).to(mdt) for _ in range(R)] if mdt is not None else None
            i = [0]

            def run():
                j = i[0] % R
                i[0] += 1
                ops.hgemm_tn(As[j], b, None if (acc or mdt is not None) else bias, relu, A2s[j] if A2s else None,
                             Ms[j] if Ms else None, out=Os[j], accumulate=acc)
            t = timeit(run)
            byt = M * K * 2 + 2 * K * N + M * N * Os[0].element_size() * (2 if acc else 1) + (M * N * 2 if Ms else 0)
            fl = 2.0 * M * N * K
            out.append(dict(kind='gemm', name=name, M=M, K=K, N=N, us=t, tflops=fl / t / 1e6, tbs=byt / t / 1e6))
            del As, A2s, Os, Ms
    return out


def bench_wgrad():
    import torch
    from transformertts_amd import ops
    dev, out = 'cuda:0', []
    R = 3
    for M in (28800, 6400):
        for name, K, N in [('w2', 1024, 256), ('w1', 256, 1024), ('wo', 256, 256), ('wqkv', 256, 768)]:
            Xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]
            Ys = [torch.randn(M, N, device=dev).bfloat16() for _ in range(R)]
            dw = torch.zeros(K, N, device=dev)
            db = torch.zeros(N, device=dev)
            i = [0]

            def run():
                j = i[0] % R
                i[0] += 1
                ops.hgemm_wgrad_rows(Xs[j], Ys[j], dw, db)
            t = timeit(run)
            byt = M * (K + N) * 2 + 4 * K * N
            out.append(dict(kind='wgrad', name=name, M=M, K=K, N=N, us=t, tflops=2.0 * M * N * K / t / 1e6, tbs=byt / t / 1e6))
    return out


def bench_attn():
    import torch
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream, check
    dev, out = 'cuda:0', []
    l = _lib.lib()
    # (.., io): 'h' = bf16 tensors (TTSMI_BF16_IO, the dense blocks), 'f' = fp32 tensors, bf16 MFMA (TTSMI_BF16, the conv blocks)
    for (B, H, T, dh, pdrop, io) in [(32, 4, 900, 64, 0.1, 'h'), (32, 4, 900, 64, 0.0, 'h'), (32, 4, 200, 64, 0.1, 'h'),
                                     (32, 2, 900, 192, 0.1, 'f'), (32, 2, 900, 192, 0.1, 'h')]:
        d = H * dh
        R = 3
        tdt = torch.bfloat16 if io == 'h' else torch.float32
        IO = _lib.TTSMI_BF16_IO if io == 'h' else _lib.TTSMI_BF16
        qkvs = [(torch.randn(B * T, 3 * d, device=dev) * 0.5).to(tdt) for _ in range(R)]
        dctx = [(torch.randn(B * T, d, device=dev) * 0.1).to(tdt) for _ in range(R)]
        pad = torch.zeros(B, T, dtype=torch.uint8, device=dev)
        klen = torch.full((B,), T, dtype=torch.int32, device=dev)
        ctx = torch.empty(B * T, d, device=dev, dtype=tdt)
        lse = torch.empty(B, H, T, device=dev)
        dqkv = torch.empty(B * T, 3 * d, device=dev, dtype=tdt)
        step = torch.zeros(1, dtype=torch.int64, device=dev)
        ws = torch.empty(int(l.ttsmi_attention_bwd_ws_bytes(B, H, T, dh)), dtype=torch.uint8, device=dev)
        i = [0]

        def fwd():
            j = i[0] % R
            i[0] += 1
            check(l.ttsmi_attention_fwd(_p(qkvs[j]), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, pdrop, 7,
                                        _p(step), 3, IO, _stream()), 'attention_fwd')

        def bwd():
            j = i[0] % R
            i[0] += 1
            check(l.ttsmi_attention_bwd(_p(qkvs[j]), _p(pad), _p(klen), _p(ctx), _p(dctx[j]), _p(lse), _p(dqkv), B, H, T,
                                        dh, pdrop, 7, _p(step), 3, _p(ws), ws.numel(), IO, _stream()),
                  'attention_bwd')
        if pdrop > 0:
            drop = ops.DropCtx(7, step)
            dm = ops.attention_dropmask(B, H, T, pdrop, drop, 3, dev)

            def gen():
                ops.attention_dropmask(B, H, T, pdrop, drop, 3, dev)

            def fwd_m():
                j = i[0] % R
                i[0] += 1
                check(l.ttsmi_attention_fwd_masked(_p(qkvs[j]), _p(pad), _p(klen), _p(ctx), _p(lse), B, H, T, dh, pdrop,
                                                   _p(dm), IO, _stream()), 'attention_fwd_masked')

            def bwd_m():
                j = i[0] % R
                i[0] += 1
                check(l.ttsmi_attention_bwd_masked(_p(qkvs[j]), _p(pad), _p(klen), _p(ctx), _p(dctx[j]), _p(lse), _p(dqkv),
                                                   B, H, T, dh, pdrop, _p(dm), _p(ws), ws.numel(), IO,
                                                   _stream()), 'attention_bwd_masked')
            fl = 4.0 * B * H * T * T * dh
            cases = [('gen bits', gen, 0), ('fwd bits', fwd_m, 1), ('bwd bits', bwd_m, 2)]
            for nm, fn, mult in cases:
                t = timeit(fn, n=20)
                out.append(dict(kind='attn', name=f'{nm} p={pdrop}' + ('' if io == 'h' else ' f32io'), M=B * T, K=T, N=dh, us=t, tflops=mult * fl / t / 1e6 + 1e-9, tbs=0.0))
        tf_ = timeit(fwd, n=20)
        tb_ = timeit(bwd, n=20)
        fl = 4.0 * B * H * T * T * dh
        out.append(dict(kind='attn', name=f'fwd p={pdrop}' + ('' if io == 'h' else ' f32io'), M=B * T, K=T, N=dh, us=tf_, tflops=fl / tf_ / 1e6, tbs=0.0))
        out.append(dict(kind='attn', name=f'bwd p={pdrop}' + ('' if io == 'h' else ' f32io'), M=B * T, K=T, N=dh, us=tb_, tflops=2 * fl / tb_ / 1e6, tbs=0.0))
    return out


def bench_ln():
    import torch
    from transformertts_amd import ops
    dev, out = 'cuda:0', []
    drop = ops.DropCtx(1, torch.zeros(1, dtype=torch.int64, device=dev))
    for M in (28800, 6400):
        C, R = 256, 3
        xs = [torch.randn(M, C, device=dev) for _ in range(R)]
        rs = [torch.randn(M, C, device=dev) for _ in range(R)]
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        pad = torch.zeros(M, dtype=torch.uint8, device=dev)
        i = [0]
        saved = {}

        def fwd():
            j = i[0] % R
            i[0] += 1
            saved['o'] = ops._ln_fwd(xs[j], rs[j], g, b, pad, 0.1, 5, drop, True)
        t = timeit(fwd)
        out.append(dict(kind='ln', name='fwd (+bf16 copy)', M=M, K=C, N=C, us=t, tflops=0.0, tbs=M * C * 14 / t / 1e6))
        y, yh, mean, rstd = saved['o']

        def bwd():
            j = i[0] % R
            i[0] += 1
            ops._ln_bwd(xs[j], xs[(j + 1) % R], rs[j], g, mean, rstd, pad, 0.1, 5, drop, dg, db, dx_bf16=True)
        t = timeit(bwd)
        out.append(dict(kind='ln', name='bwd (bf16 dx)', M=M, K=C, N=C, us=t, tflops=0.0, tbs=M * C * 18 / t / 1e6))
    return out


def bench_rowgemm():
    import torch
    from transformertts_amd import _lib, ops
    from transformertts_amd.ops import _p, _stream, check
    l = _lib.lib()
    dev, out = 'cuda:0', []
    N, R = 256, 3
    PD = float(os.environ.get('KBENCH_PDROP', '0.1'))          # dropout rate of the fused res-norms (0 = no hash)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    drop = ops.DropCtx(7, step)
    for M in (28800, 6400):
        pad = torch.zeros(M, dtype=torch.uint8, device=dev)
        gam, bet, bias = torch.ones(N, device=dev), torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        for name, K in (('o+LN1 fwd', 512), ('ffn2+LN2 fwd', 1024)):
            As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]
            res = [torch.randn(M, N, device=dev) for _ in range(R)]
            wt = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            y = torch.empty(M, N, device=dev); yh = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            xh = torch.empty(M, N, device=dev, dtype=torch.bfloat16); rstd = torch.empty(M, device=dev)
            o = torch.empty(M, N, device=dev)
            i = [0]

            def fused():
                j = i[0] % R; i[0] += 1
                check(l.ttsmi_hgemm_ln_fwd(_p(As[j]), K, None, 0, 0, _p(wt), K, _p(bias), _p(res[j]), _p(gam), _p(bet), _p(pad),
                                           PD, 5, 7, _p(step), 1e-6, _p(y), _p(yh), _p(xh), _p(rstd), M, N, K, _stream()))

            def unfused():
                j = i[0] % R; i[0] += 1
                ops.hgemm_tn(As[j], wt, bias, out=o)
                ops._ln_fwd(o, res[j], gam, bet, pad, PD, 5, drop, True)
            resh = [r.bfloat16() for r in res]

            def fused_h():               # the form the planned blocks launch: bf16 residual, no fp32 output
                j = i[0] % R; i[0] += 1
                check(l.ttsmi_hgemm_ln_fwd_h(_p(As[j]), K, None, 0, 0, _p(wt), K, _p(bias), _p(resh[j]), _p(gam), _p(bet), _p(pad),
                                             PD, 5, 7, _p(step), 1e-6, None, _p(yh), _p(xh), _p(rstd), M, N, K, _stream()))
            for nm, fn in ((name + ' fused', fused), (name + ' fused res16', fused_h), (name + ' gemm+ln', unfused)):
                t = timeit(fn)
                out.append(dict(kind='rowg', name=nm, M=M, K=K, N=N, us=t, tflops=2.0 * M * N * K / t / 1e6, tbs=0.0))
        K = 1024
        As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(R)]
        part = [torch.randn(M, N, device=dev) for _ in range(R)]
        wb = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        xh = torch.randn(M, N, device=dev).bfloat16(); rstd = torch.ones(M, device=dev)
        dxb = torch.empty(M, N, device=dev, dtype=torch.bfloat16); dres = torch.empty(M, N, device=dev)
        ws1 = torch.empty(int(l.ttsmi_layernorm_partials_bytes(l.ttsmi_hgemm_ln_bwd_nparts(M), N)), dtype=torch.uint8, device=dev)
        ws2 = torch.empty(int(l.ttsmi_layernorm_partials_bytes(l.ttsmi_layernorm_bwd_xhat_nparts(M), N)), dtype=torch.uint8, device=dev)
        o = torch.randn(M, N, device=dev); mean = torch.zeros(M, device=dev)
        dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
        i = [0]

        def fusedb():
            j = i[0] % R; i[0] += 1
            check(l.ttsmi_hgemm_ln_bwd(_p(As[j]), K, _p(wb), K, _p(part[j]), _p(xh), _p(rstd), _p(gam), _p(pad), PD, 5, 7,
                                       _p(step), _p(dxb), _p(dres), _p(ws1), ws1.numel(), M, N, K, _stream()))

        def unfusedb():
            j = i[0] % R; i[0] += 1
            ops.hgemm_tn(As[j], wb, out=part[j], accumulate=True)
            ops._ln_bwd(part[j], o, part[(j + 1) % R], gam, mean, rstd, pad, PD, 5, drop, dg, db, dx_bf16=True)

        def xhatb():
            j = i[0] % R; i[0] += 1
            check(l.ttsmi_layernorm_bwd_xhat(_p(part[j]), _p(xh), _p(rstd), _p(gam), _p(pad), PD, 5, 7, _p(step), _p(dxb),
                                             _p(dres), _p(ws2), ws2.numel(), M, N, _stream()))
        parth = [q.bfloat16() for q in part]
        dresh = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def fusedb_h():                  # bf16 residual gradients in and out (chained blocks)
            j = i[0] % R; i[0] += 1
            check(l.ttsmi_hgemm_ln_bwd_dual_h(_p(As[j]), K, None, 0, 0, _p(wb), K, None, 0, _p(parth[j]), _p(xh), _p(rstd), _p(gam),
                                              _p(pad), PD, 5, 7, _p(step), _p(dxb), _p(dresh), 1, _p(ws1), ws1.numel(), M, N, K,
                                              _stream()))
        for nm, fn in (('da+LN1 bwd fused', fusedb), ('da+LN1 bwd fused res16', fusedb_h), ('da+LN1 bwd gemm+ln', unfusedb),
                       ('LN bwd xhat', xhatb)):
            t = timeit(fn)
            out.append(dict(kind='rowg', name=nm, M=M, K=K, N=N, us=t, tflops=2.0 * M * N * K / t / 1e6, tbs=0.0))
    return out


def run_all(only):
    res = []
    if only in (None, 'gemm'):
        res += bench_gemm()
    if only == 'gemm-small':                 # inference (batch 1: 2304 frames / 400 phonemes) and encoder-side rows
        res += bench_gemm((6400, 2304, 400))
    if only in (None, 'wgrad'):
        res += bench_wgrad()
    if only in (None, 'attn'):
        res += bench_attn()
    if only in (None, 'ln'):
        res += bench_ln()
    if only in (None, 'rowgemm'):
        res += bench_rowgemm()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', nargs='*', default=None, help='e.g. base TTSMI_HGEMM_OCC4=1 "A=1,B=2"')
    ap.add_argument('--only', default=None)
    ap.add_argument('--json', default=None, help='append the table as JSON lines to this file')
    ap.add_argument('--child', action='store_true')
    args = ap.parse_args()
    if args.child or not args.variants:
        res = run_all(args.only)
        if args.child:
            print('KBENCH ' + json.dumps(res))
        else:
            show({'default': res})
        return
    tables = {}
    for v in args.variants:
        env = dict(os.environ)
        if v != 'base':
            for kv in v.split(','):
                k, val = kv.split('=')
                env[k] = val
        cmd = [sys.executable, os.path.abspath(__file__), '--child'] + (['--only', args.only] if args.only else [])
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('KBENCH ')]
        if r.returncode != 0 or not line:
            print(f'variant {v} failed rc={r.returncode}\n{r.stderr[-2000:]}')
            continue
        tables[v] = json.loads(line[0][7:])
    show(tables)
    if args.json:
        with open(args.json, 'a') as f:
            for v, t in tables.items():
                f.write(json.dumps({'variant': v, 'rows': t}) + '\n')


def show(tables):
    names = list(tables)
    if not names:
        return
    rows = tables[names[0]]
    print(f'{"kernel":6s} {"case":18s} {"M":>6s} {"K":>5s} {"N":>5s} | ' + ' | '.join(f'{n[-22:]:>22s}' for n in names))
    for i, r in enumerate(rows):
        cells = []
        for n in names:
            t = tables[n][i] if i < len(tables[n]) else None
            if t is None:
                cells.append(' ' * 22)
            elif t['tflops']:
                cells.append(f'{t["us"]:8.1f}us {t["tflops"]:6.0f}TF {t["tbs"]:4.1f}')
            else:
                cells.append(f'{t["us"]:8.1f}us {t["tbs"]:6.2f}TB/s   ')
        print(f'{r["kind"]:6s} {r["name"]:18s} {r["M"]:6d} {r["K"]:5d} {r["N"]:5d} | ' + ' | '.join(cells))


if __name__ == '__main__':
    main()
