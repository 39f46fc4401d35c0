#!/usr/bin/env python
"""bench.py - the headline measurement of BASELINE.json: mel-frames/sec of a full ForwardTransformer
train step (forward + backward + TF-form Adam [+ gradient all-reduce]) on synthetic
LJSpeech-shaped batches, 1/2/4/8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload = "configs[1]"): d_model 256, 6+6 dense blocks, 4 heads, FFN 1024, 80-bin
mel, predictors [256,226] k=3, per-GPU batch 32, every sample 200 phonemes / 900 mel frames
(SURVEY.md section 8d max-shape set), dropout 0.1 as in the reference's training config.  Weak
scaling: per-GPU batch fixed, global batch = 32*N, the flat fp32 gradient buffer is averaged with two
RCCL all-reduces per step (decoder half overlapped with the encoder's backward).  Inputs are resident in HBM before the timed region.

One JSON line is printed by rank 0.  Besides the driver contract it carries
  roofline     - the dominant kernel family of the step (the one with the most launch time): its
                 algorithmic bytes and FLOPs per launch / its average launch
                 duration, measured live with HIP events on the stream each launch goes to, in one
                 extra instrumented step after the timed region.  `bound` is chosen by the family's
                 arithmetic intensity against the ridge (peak FLOP/s / peak HBM B/s): the bf16 path's
                 K=256..1024 GEMMs sit below it (HBM-bound), the exact-fp32 path's above (MFMA-bound).
                 `traffic` = measured HBM bytes per launch from the committed PMC passes;
  cpu_baseline - the reference-restatement oracle (torch-CPU fp32, same graph, attention maps
                 materialised like the reference) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
# The roof that binds the bf16 flash-attention kernels is neither of the two above: the softmax / keep-bit / conversion
# arithmetic runs on the vector ALU of the SIMD whose matrix pipe does the products.  Vector-ALU issue cycles per SCORE
# ROW (one wave64 instruction stream covering 64 scores), keep-bit dropout, dh = 64: the instruction mix of each kernel's
# inner loop read from its ISA (forward ~6 per score; dQ ~8; dK/dV: 2 v_and + v_bfe + v_sub + 1.2 v_fma + v_exp +
# v_cvt_pk + 0.9 packed = 8.2) times the measured issue cost per wave64 instruction at 4 waves per SIMD
# (profiles/r04_valu_rates.txt: v_fma / v_and 2.9, v_cndmask / v_cvt_pk / v_bfe 4.25, packed fp32 4.35, v_exp_f32 8.2).
ATTN_VALU_CYCLES_PER_SCORE_ROW = {'fwd': 26.9, 'bwd': 30.0 + 32.7}           # bwd = dQ + dK/dV (two recomputations)
SIMDS, SIMD_CLOCK_GHZ = 1024, 2.3                                            # 256 CUs x 4; clock read in the probes: 2.2-2.4


def add_attention_valu_roofs(per_kernel: dict, cfg: dict) -> None:
    """valu_roof_* fields for the bf16 attention families of a `roofline.per_kernel` table (dh = 64 architectures: the
    instruction counts above are those kernels'); never raises - a missing family or key just leaves the table as it is."""
    try:
        heads = cfg.get('decoder_num_heads') or []
        dh = cfg['decoder_model_dimension'] // heads[0] if heads else 0
        if dh != 64:
            return
        for fam, which in ((HATTN_FWD, 'fwd'), (HATTN_BWD, 'bwd')):
            e = per_kernel.get(fam)
            if e and e.get('gflop') and e.get('ms'):
                e.update(attention_valu_roof(e['gflop'] * 1e9, e['ms'], dh, which))
    except Exception:       # noqa: BLE001 - a reporting extra must not cost the bench line
        pass


def attention_valu_roof(flops: float, ms: float, dh: int, which: str) -> dict:
    """Vector-ALU roof of an attention family: `flops` = its algorithmic FLOPs (forward 4 T^2 dh, backward 8 T^2 dh per
    head), `ms` its measured launch time.  valu_roof_ms = the time the chip's vector ALUs alone need for the scores'
    arithmetic; valu_roof_frac = that / ms (1.0 = the kernel runs at the VALU roof)."""
    scores = flops / ((4.0 if which == 'fwd' else 8.0) * dh)
    cyc = ATTN_VALU_CYCLES_PER_SCORE_ROW[which]
    roof_ms = scores / 64.0 * cyc / SIMDS / (SIMD_CLOCK_GHZ * 1e9) * 1e3
    return {'valu_cycles_per_score_row': cyc, 'valu_roof_ms': roof_ms, 'valu_roof_frac': (roof_ms / ms if ms else None)}


def workload_config(name: str):
    from transformertts_amd.utils.synthetic import make_config
    if name == 'configs[1]':
        return make_config(), dict(B=32, Tp=200, Tm=900)
    if name == 'ref-default':
        # the reference's shipped configuration (config/training_config.yaml:104-118): d_model 384, 2 heads (dh = 192),
        # 6+6 self-attention CONV blocks with filters [1536, 384], kernel 3 - the architecture of the released weights
        return make_config(d_model=384, enc_heads=(2,) * 6, dec_heads=(2,) * 6, ffn=None, enc_dense_blocks=0,
                           dec_dense_blocks=0, conv_filters=(1536, 384)), dict(B=32, Tp=200, Tm=900)
    if name == 'configs[0]':
        return make_config(d_model=64, enc_heads=(2, 2), dec_heads=(2, 2), ffn=256, dur_filters=(64, 64),
                           pitch_filters=(64, 64)), dict(B=4, Tp=50, Tm=200)
    raise ValueError(name)


# ------------------------------------------------------------------------------------------------
# per-entry-point algorithmic FLOPs (2 per MAC, no recompute) from the C-ABI arguments
# ------------------------------------------------------------------------------------------------
def _flops(name, a):
    if name == 'ttsmi_linear_fwd':
        return 2.0 * a[10] * a[11] * a[12]
    if name == 'ttsmi_linear_dgrad':
        return 2.0 * a[8] * a[9] * a[10]
    if name == 'ttsmi_linear_wgrad':
        return 2.0 * a[7] * a[8] * a[9]
    if name in ('ttsmi_conv1d_fwd',):
        return 2.0 * a[4] * a[5] * a[6] * a[7] * a[8]
    if name == 'ttsmi_conv1d_dgrad':
        return 2.0 * a[4] * a[5] * a[6] * a[7] * a[8]
    if name == 'ttsmi_conv1d_wgrad':
        return 2.0 * a[4] * a[5] * a[6] * a[7] * a[8]
    if name == 'ttsmi_hgemm_tn':
        return 2.0 * a[13] * a[14] * a[15]
    if name == 'ttsmi_hgemm_wgrad':
        return 2.0 * a[6] * a[7] * a[8]
    if name == 'ttsmi_hgemm_wgrad_rows':
        return 2.0 * a[9] * a[10] * a[11]
    if name == 'ttsmi_attention_fwd':
        B, H, T, dh = a[5], a[6], a[7], a[8]
        return 4.0 * B * H * T * T * dh                      # QK^T + PV
    if name == 'ttsmi_attention_bwd':
        B, H, T, dh = a[7], a[8], a[9], a[10]
        return 8.0 * B * H * T * T * dh                      # dV, dP, dQ, dK (S recompute not counted)
    return 0.0


def _bytes(name, a):
    """Algorithmic HBM bytes of one launch: every operand read once, every result written once (an
    accumulate epilogue also reads its output), weights included; no re-reads, no workspace."""
    if name == 'ttsmi_hgemm_tn':
        a_f32, K1, relu_src, M, N, K, flags, taps = a[1], a[5], a[9], a[13], a[14], a[15], a[16], a[17]
        k_cols = K // taps if taps > 1 else K                 # implicit-GEMM conv reads x once, not k times
        out_b = 2 if flags & 4 else 4
        byt = M * k_cols * (4 if a_f32 else 2) + 2.0 * K * N + M * N * out_b * (2 if flags & 2 else 1)
        if relu_src:
            byt += M * N * (2 if flags & 8 else 4)
        return byt
    if name == 'ttsmi_hgemm_wgrad_rows':
        xh, yh, rows, kin, n, taps = a[1], a[4], a[9], a[10], a[11], a[12]
        k_cols = kin // taps if taps > 1 else kin
        return rows * k_cols * (2 if xh else 4) + rows * n * (2 if yh else 4) + 4.0 * kin * n
    if name == 'ttsmi_hgemm_wgrad':
        rows, kin, n = a[6], a[7], a[8]
        return 2.0 * rows * (kin + n) + 4.0 * kin * n
    if name == 'ttsmi_attention_fwd':
        B, H, T, dh, dtype = a[5], a[6], a[7], a[8], a[13]
        rows, d = B * T, H * dh
        return rows * 3 * d * (2 if dtype == 2 else 4) + rows * d * 4 + B * H * T * 4
    if name == 'ttsmi_attention_bwd':
        B, H, T, dh, dtype = a[7], a[8], a[9], a[10], a[17]
        rows, d = B * T, H * dh
        qb = 2 if dtype == 2 else 4
        # dq pass: q,k,v + ctx + dctx + lse -> dq, delta;  dkv pass: q,k,v + dctx + lse + delta -> dk,dv
        return (2 * rows * 3 * d * qb + rows * d * 4 * 3 + rows * 3 * d * qb + 4.0 * B * H * T * 4)
    if name == 'ttsmi_linear_fwd':
        M, N, K = a[10], a[11], a[12]
        return 4.0 * (M * K + K * N + M * N)
    if name == 'ttsmi_linear_dgrad':
        M, N, K, acc = a[8], a[9], a[10], a[11]
        return 4.0 * (M * N + K * N + M * K * (2 if acc else 1)) + (4.0 * M * K if a[4] else 0)
    if name == 'ttsmi_linear_wgrad':
        M, N, K = a[7], a[8], a[9]
        return 4.0 * (M * K + M * N + K * N)
    if name in ('ttsmi_conv1d_fwd', 'ttsmi_conv1d_dgrad', 'ttsmi_conv1d_wgrad'):
        B, T, Cin, Cout, k = a[4], a[5], a[6], a[7], a[8]
        return 4.0 * (B * T * (Cin + Cout) + k * Cin * Cout)
    if name == 'ttsmi_add_layernorm_fwd':
        M, C = a[18], a[19]
        return 4.0 * M * C * (2 + (1 if a[1] else 0))
    if name == 'ttsmi_add_layernorm_bwd':
        M, C = a[22], a[23]
        return 4.0 * M * C * (3 + (1 if a[2] else 0) + (1 if a[18] and a[18] != a[17] else 0))
    if name == 'ttsmi_adam_tf':
        return a[4] * (7 * 4.0 + (2 if a[10] else 0))
    return 0.0


KERNEL_OF = {
    'ttsmi_linear_fwd': 'gemm_f32_kernel<A_KC,B_NC> (Dense/Conv1D forward)',
    'ttsmi_conv1d_fwd': 'gemm_f32_kernel<A_KC,B_NC> (Dense/Conv1D forward)',
    'ttsmi_linear_dgrad': 'gemm_f32_kernel<A_KC,B_KC> (dgrad)',
    'ttsmi_conv1d_dgrad': 'gemm_f32_kernel<A_KC,B_KC> (dgrad)',
    'ttsmi_linear_wgrad': 'gemm_f32_kernel<A_MC,B_NC> (wgrad, + split reduce + bias colsum)',
    'ttsmi_conv1d_wgrad': 'gemm_f32_kernel<A_MC,B_NC> (wgrad, + split reduce + bias colsum)',
    'ttsmi_hgemm_tn': 'gemm_bf16_kernel / gemm_bf16_dma_kernel / gemm_bf16_deep_kernel / gemm_k256_kernel (Dense/Conv1D forward + dgrad, bf16 MFMA)',
    'ttsmi_hgemm_wgrad': 'gemm_bf16_kernel<A=bf16> (wgrad, bf16 MFMA, + split reduce)',
    'ttsmi_hgemm_wgrad_rows': 'wgrad_dma_kernel / wgrad_rows_kernel (wgrad from row-major activations, bf16 MFMA, + split reduce)',
    'ttsmi_attention_fwd': 'attn_fwd_kernel (exact fp32 MFMA)',
    'ttsmi_attention_bwd': 'attn_bwd_dq_kernel + attn_bwd_dkv_kernel (exact fp32 MFMA)',
}
RIDERS = 'hbm-bound riders (LN, lenreg, loss, Adam, ...)'
SIDE = ' [side stream]'
HATTN_FWD = 'hattn_fwd_kernel (bf16 MFMA flash attention forward)'
HATTN_BWD = 'hattn_bwd_dq_kernel + hattn_bwd_dkv_kernel (bf16 MFMA flash attention backward)'
ROWGEMM = 'rowgemm_dma_kernel / rowgemm_kernel (full-row GEMM + fused LayerNorm forward / backward, bf16 MFMA)'
CHAIN = 'dense_chain16_kernel / dense_chain_kernel (row-local chain of a dense block: o-projection + res-norm 1 + FFN + res-norm 2 + next qkv in one launch - and its backward between the two res-norms, bf16 MFMA)'
# launch groups announced by the C++ block launcher (ttsmi_set_launch_observer) -> kernel family
OBSERVED = {'ttsmi_hgemm_tn': KERNEL_OF['ttsmi_hgemm_tn'], 'ttsmi_hgemm_ln_fwd': ROWGEMM, 'ttsmi_hgemm_ln_bwd': ROWGEMM,
            'ttsmi_attention_fwd': HATTN_FWD, 'ttsmi_attention_bwd': HATTN_BWD, 'ttsmi_dense_chain_fwd': CHAIN, 'ttsmi_dense_chain_bwd': CHAIN,
            'ttsmi_hgemm_wgrad_rows': KERNEL_OF['ttsmi_hgemm_wgrad_rows']}


def kernel_family(name, args):
    if name == 'ttsmi_attention_fwd' and args[13] != 0:
        return HATTN_FWD
    if name == 'ttsmi_attention_bwd' and args[17] != 0:
        return HATTN_BWD
    return KERNEL_OF.get(name, RIDERS)


PMC_FILE = 'r06_pmc_hbm_traffic_bf16.json'
PMC_KERNELS = {       # kernel family -> (rocprof names of its kernels, names of helper kernels of the same entry point)
    KERNEL_OF['ttsmi_hgemm_tn']: (['gemm_bf16_kernel', 'gemm_bf16_dma_kernel', 'gemm_bf16_deep_kernel', 'gemm_k256_kernel'], []),
    ROWGEMM: (['rowgemm_dma_kernel', 'rowgemm_kernel'], []),
    CHAIN: (['dense_chain16_kernel', 'dense_chain16_bwd_kernel', 'dense_chain_kernel'], ['dense_chain16_pack_kernel', 'dense_chain_pack_kernel']),
    KERNEL_OF['ttsmi_hgemm_wgrad_rows']: (['wgrad_rows_kernel', 'wgrad_dma_kernel'], ['hsplit_reduce']),
    HATTN_FWD: (['hattn_fwd_kernel'], []),
    HATTN_BWD: (['hattn_bwd_dq_kernel'], ['hattn_bwd_dkv_kernel']),
}


def peak_of(kernel_name: str) -> float:
    return PEAK_BF16_MFMA_TFLOPS if 'bf16' in kernel_name else PEAK_F32_MFMA_TFLOPS


def instrumented_step(step_fn):
    """Run one step with every C-ABI call bracketed by HIP events on the stream it launches on.
    Returns per-call records (entry point, algorithmic flops, algorithmic bytes, shape key, ms)."""
    from transformertts_amd import _lib
    recs = []
    main_stream = torch.cuda.current_stream().cuda_stream

    ext = {}

    def hook(name, args, fn):
        if (name.endswith('_bytes') or name.endswith('_nparts') or
                name in ('ttsmi_last_error', 'ttsmi_version', 'ttsmi_last_kernel', 'ttsmi_dense_block_fwd', 'ttsmi_dense_block_bwd',
                         'ttsmi_dense_stack_fwd', 'ttsmi_dense_stack_bwd', 'ttsmi_set_launch_observer', 'ttsmi_dense_block_bwd_chained', 'ttsmi_debug_stream_create_cu_mask', 'ttsmi_ft_train_step') or
                'comm' in name or name.endswith('_supported')):
            return fn(*args)          # queries / entry points without a stream; the block and stack launchers announce
            #                           their launches through the observer
        # the launch stream is the entry point's last argument (the weight gradients pass the side
        # stream's handle explicitly while torch's current stream stays the main one)
        h = args[-1] if isinstance(args[-1], int) else None
        side = h is not None and h != main_stream
        st = None
        if side:
            st = ext.get(h)
            if st is None:
                st = ext[h] = torch.cuda.ExternalStream(h)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st) if st is not None else e0.record()
        rc = fn(*args)
        e1.record(st) if st is not None else e1.record()
        fam = kernel_family(name, args)
        if side or torch.cuda.current_stream().cuda_stream != main_stream:
            fam += SIDE           # weight-gradient launches on the second HIP stream overlap the main stream
        key = tuple(x for x in args if isinstance(x, int) and not isinstance(x, bool) and 0 <= x < (1 << 26))
        recs.append((fam, name, _flops(name, args), _bytes(name, args), key, e0, e1))
        return rc

    # launches issued from C++ (ttsmi_dense_block_fwd / _bwd) announce themselves through the library's observer hook
    open_ev = {}

    def observe(phase, name, flops, byt, stream):
        name = name.decode()
        h = stream or 0
        side = h != main_stream
        st = None
        if side:
            st = ext.get(h)
            if st is None:
                st = ext[h] = torch.cuda.ExternalStream(h)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(st) if st is not None else ev.record()
        if phase == 0:
            open_ev[(name, h)] = ev
            return
        fam = OBSERVED.get(name, RIDERS) + (SIDE if side else '')
        recs.append((fam, name, flops, byt, (int(flops), int(byt)), open_ev.pop((name, h)), ev))

    cb = _lib.LAUNCH_OBSERVER(observe)
    _lib.set_trace(hook)
    _lib.lib()._cdll.ttsmi_set_launch_observer(cb)
    try:
        step_fn()
        torch.cuda.synchronize()
    finally:
        _lib.set_trace(None)
        _lib.lib()._cdll.ttsmi_set_launch_observer(None)
    return [(fam, n, fl, by, key, e0.elapsed_time(e1)) for fam, n, fl, by, key, e0, e1 in recs]


def group_records(recs):
    """kernel family -> [launches, flops, bytes, ms]"""
    groups = {}
    for k, _name, fl, by, _key, ms in recs:
        g = groups.setdefault(k, [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += fl
        g[2] += by
        g[3] += ms
    return groups


def _pmc_file(name):
    """A committed PMC summary, or None when it is absent or was NOT collected on the build in this tree: the file is
    stamped with transformertts_amd.build.library_digest() by tools/rocpd_pmc_traffic.py, and a kernel change without a
    PMC refresh must not report stale bytes (round-3 review)."""
    path = os.path.join(ROOT, 'profiles', name)
    try:
        with open(path) as f:
            d = json.load(f)
        from transformertts_amd.build import library_digest
        return d if d.get('lib_digest') == library_digest() else None
    except (OSError, ValueError):
        return None


def stamped_pmc(name, prefixes):
    d = _pmc_file(name)
    if d is None:
        return None
    hit = [v for k, v in d.get('kernels', {}).items() if any(k.startswith(p) for p in prefixes)]
    return hit[0]['hbm_bytes_per_launch'] if hit else None


def pmc_traffic(kernel_family: str):
    """HBM bytes per launch of a kernel family from the committed PMC passes (profiles/, collected by
    tools/gpu_profile.sh with separate FETCH_SIZE / WRITE_SIZE runs and the guide's gfx950 correction)."""
    entry = PMC_KERNELS.get(kernel_family.replace(SIDE, ''))
    d = _pmc_file(PMC_FILE)
    if not entry or d is None:
        return None
    mains, helpers = entry
    ks = d['kernels']
    n = b = 0.0
    for k, v in ks.items():
        if any(k.startswith(p) for p in mains):
            n += v['launches']
            b += v['launches'] * v['hbm_bytes_per_launch']
        elif any(k.startswith(p) for p in helpers):   # helper kernels add bytes, not launches
            b += v['launches'] * v['hbm_bytes_per_launch']
    return b / n if n else None


def pmc_step_totals():
    """Whole-step HBM bytes and kernel launches from the same committed PMC passes (None when absent / stale)."""
    d = _pmc_file(PMC_FILE)
    if d is None or 'hbm_bytes_per_step' not in d:
        return None
    return {'hbm_gb_per_step': d['hbm_bytes_per_step'] / 1e9, 'kernel_launches_per_step': d['launches_per_step'],
            'lib_digest': d.get('lib_digest'),
            'source': 'profiles/' + PMC_FILE + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench on the '
                      'build with this digest, all kernels of a step, both streams)'}


def usable_cpus() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup v2 cpu.max quota).  The GPU
    boxes expose 256 logical CPUs but cap the container at a 16-CPU quota; running an OpenMP pool
    wider than the quota makes the CPU leg orders of magnitude slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(cfg, shape, threads):
    """Reference-restatement CPU baseline (TF2 is not installable offline): oracle/ft_oracle.py in
    torch-CPU fp32, same graph incl. materialised attention maps and dropout, on a bounded sample
    (a slice of the batch axis of the same workload)."""
    from oracle import ft_oracle as fo
    torch.set_num_threads(threads)
    Bs = max(1, min(4, shape['B']))
    W = fo.init_weights(cfg, seed=0)
    m = fo.ForwardTransformerOracle(cfg, W, torch.float32)
    batch = fo.synthetic_batch(Bs, shape['Tp'], shape['Tm'], seed=1234)
    m.train_step(*batch)                                  # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < 12.0 and n < 40):     # a bounded sample: >= 3 steps, ~12 s of CPU work
        m.train_step(*batch)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {'value': Bs * shape['Tm'] / dt, 'unit': 'mel-frames/s', 'cores': threads, 'kind': 'port',
            # a bounded SAMPLE of the workload (SURVEY 8d names 1 + 3 steps at B = 32: ~40 s of CPU per step count here would
            # take the default run past a minute of CPU legs): samples are independent, so frames/s of B = 4 is the rate
            'sample_batch': Bs, 'headline_batch': shape['B'],
            'sample': f'B={Bs} samples of the same {shape["Tp"]}-phoneme/{shape["Tm"]}-frame workload, '
                      f'1 warm-up + {n} timed train steps ({n * dt:.1f} s), torch-CPU fp32 restatement of the TF2 '
                      f'graph (TF2 unavailable offline), {threads} threads, {dt:.2f} s/step; like the reference it '
                      f'materialises and returns the 12 attention maps every step (the GPU `value` does not: compare '
                      f'with ms_per_step_with_attention_maps)'}


def predict_cases(precision, reps, graphs=(True, False), with_maps=True):
    """Latency cases of ForwardTransformer.predict at 400 phonemes with forced durations (mean 5.7 frames per phoneme):
    batch 1 and batch 64, hipGraph-captured and/or eager, optionally with the 12 attention maps at batch 1."""
    from transformertts_amd.model.models import ForwardTransformer
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    cfg, _ = workload_config('configs[1]')
    rng = np.random.default_rng(1234)
    Tp = 400
    cases = []
    n_par = 0
    for graph in graphs:
        model = ForwardTransformer.from_config(dict(cfg, device=str(dev), seed=0, precision=precision,
                                                    graph_inference=graph))
        for B in (1, 64):
            tok = torch.from_numpy(rng.integers(1, 127, size=(B, Tp)).astype(np.int32)).to(dev)
            dur = torch.from_numpy(rng.multinomial(int(5.7 * Tp), np.ones(Tp) / Tp, size=B).astype(np.int32)).to(dev)
            for maps in (False, True):
                if maps and (B > 1 or not graph or not with_maps):
                    continue              # 12 maps of [64, 4, 2280, 2280] fp32 are 63 GB: batch 1 only
                model.return_attention = maps
                fn = lambda: model.predict(tok, encode=False, phoneme_durations=dur)   # noqa: E731
                for _ in range(4):
                    o = fn()
                torch.cuda.synchronize()
                lat = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    o = fn()
                    torch.cuda.synchronize()
                    lat.append(time.perf_counter() - t0)
                frames = int(o['expanded_lengths'].sum().item())
                audio_s = frames * 256 / 22050.0
                lat = np.sort(np.array(lat))
                p50, p90 = float(lat[len(lat) // 2]), float(lat[int(len(lat) * 0.9)])
                cases.append({'batch': B, 'hipgraph': graph, 'attention_maps': maps, 'frames': frames,
                              'audio_seconds': audio_s, 'p50_ms': p50 * 1e3, 'p90_ms': p90 * 1e3, 'rtf_p50': p50 / audio_s,
                              'mel_frames_per_s': frames / p50})
        n_par = int(model.params.n_params)
        del model
    return cfg, Tp, cases, n_par


def predict_roofline(cfg, Tp, head, big, n_par):
    """What bounds predict.  Batch 1 is a chain of ~100 graph nodes over 2 280 rows: its algorithmic bytes (every bf16
    weight once + the activations of the path at their stored widths) against the HBM roof say how far from memory-bound
    a single utterance is (latency-bound); batch 64 is the same graph on 146 k rows, priced against the bf16 MFMA roof."""
    d, F, L = cfg['decoder_model_dimension'], cfg['decoder_feed_forward_dimension'], len(cfg['decoder_num_heads'])

    def fwd_flops(rows_enc, rows_dec, B, Tm):
        per_row = 2.0 * (3 * d * d + 2 * d * d + 2 * d * F)                       # qkv, [h | ctx] Wo, the two FFN layers
        attn = lambda rows, T: 4.0 * rows * T * d                                  # QK^T and PV over all heads
        return L * (per_row * (rows_enc + rows_dec) + attn(rows_enc, Tp) + attn(rows_dec, Tm)) + 2.0 * rows_dec * d * 80
    rows_dec1 = head['frames']
    act_bytes = lambda rows: L * rows * (3 * d + d + d + F + d + d + d) * 2.0 + rows * d * 4.0      # bf16 block tensors + fp32 ends
    by1 = 2.0 * n_par + act_bytes(Tp) + act_bytes(rows_dec1) + rows_dec1 * 80 * 4.0
    gbs1 = by1 / (head['p50_ms'] * 1e-3) / 1e9
    fl64 = fwd_flops(64 * Tp, big['frames'], 64, big['frames'] // 64)
    return {'bound': 'hbm', 'kernel': 'the whole batch-1 predict graph (two hipGraph replays, ~100 kernel nodes)',
            'achieved': gbs1, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs1 / 8000.0, 'traffic': None,
            'algorithmic_mb': by1 / 1e6,
            'note': 'latency-bound: ~9 us per graph node at 2 280 rows; batch 64 amortises it',
            'batch64': {'bound': 'mfma', 'achieved': fl64 / (big['p50_ms'] * 1e-3) / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s',
                        'frac': fl64 / (big['p50_ms'] * 1e-3) / 1e12 / 2500.0, 'algorithmic_gflop': fl64 / 1e9}}


def predict_bench(args):
    """BASELINE.json configs[4]: inference-only ForwardTransformer.predict, batch 1 and batch 64, long sentences (400
    phonemes), hipGraph-captured (model.graph_inference), 1 GPU: p50 / p90 latency and RTF = latency / seconds of audio
    (hop 256 @ 22.05 kHz).  Durations are forced (synthetic, mean 5.7 frames per phoneme -> ~2 280 frames) so that the
    decoder length is realistic with random-init weights.  Two lines per batch size: without the attention maps and -
    as the reference's predict returns them - with all 12 maps materialised."""
    reps = max(20, args.steps * 3)
    cfg, Tp, cases, n_par = predict_cases(args.precision, reps)
    head = next(c for c in cases if c['batch'] == 1 and c['hipgraph'] and not c['attention_maps'])
    big = next(c for c in cases if c['batch'] == 64 and c['hipgraph'] and not c['attention_maps'])
    roof = predict_roofline(cfg, Tp, head, big, n_par)
    result = {'metric': 'predict p50 latency, batch 1, 400 phonemes, hipGraph-captured', 'value': head['p50_ms'],
              'unit': 'ms', 'n_gpus': 1, 'steps': reps, 'warmup': 4, 'higher_is_better': False,
              'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
              'config': {'workload': 'BASELINE.json configs[4]: ForwardTransformer.predict, d_model=256 6+6 dense '
                                     'blocks, 400 phonemes, forced durations (mean 5.7 frames)'},
              'rtf_p50': head['rtf_p50'], 'roofline': roof, 'cases': cases}
    if not args.no_cpu_baseline:
        result['cpu_baseline'] = predict_cpu_baseline(cfg, Tp, usable_cpus())
    print(json.dumps(result))


def predict_cpu_baseline(cfg, Tp, threads, seconds=10.0):
    """The same batch-1 predict on the host: oracle/ft_oracle.py (torch-CPU fp32 restatement of model/models.py:518-550 with
    forced durations), a bounded sample of calls."""
    from oracle import ft_oracle as fo
    torch.set_num_threads(threads)
    ocfg = {k: v for k, v in cfg.items() if k not in ('device', 'seed', 'precision', 'use_graph')}
    m = fo.ForwardTransformerOracle(ocfg, fo.init_weights(ocfg, seed=0), torch.float32)
    rng = np.random.default_rng(1234)
    tok = rng.integers(1, 127, size=(1, Tp)).astype(np.int32)
    dur = rng.multinomial(int(5.7 * Tp), np.ones(Tp) / Tp, size=1).astype(np.int32)[..., None]
    with torch.no_grad():
        m.call(tok, target_durations=dur, training=False)                       # warm-up
        n, t0 = 0, time.perf_counter()
        while n < 3 or (time.perf_counter() - t0 < seconds and n < 40):
            m.call(tok, target_durations=dur, training=False)
            n += 1
    dt = (time.perf_counter() - t0) / n
    return {'value': dt * 1e3, 'unit': 'ms', 'cores': threads, 'kind': 'port',
            'sample': f'{n} batch-1 predict calls ({Tp} phonemes -> {int(dur.sum())} frames, forced durations, attention maps '
                      f'materialised as the reference returns them), torch-CPU fp32 restatement, {threads} threads'}


MEL_PMC_FILE = 'r06_pmc_hbm_traffic_mel.json'


def _mel_cpu_clip(args):
    """One clip through the NumPy restatement of data/audio.py:81-92 (pool worker: one process per clip, like the
    reference's p_uimap over wav files in create_training_data.py:63)."""
    from oracle import mel_oracle as mo
    return mo.mel_spectrogram(args).shape[0]


def mel_cpu_baseline(clips, bytes_per_clip, cores, single_seconds=3.0, pool_seconds=10.0):
    """NumPy restatement (oracle/mel_oracle.py: rfft + dense mel matmul, what librosa does) timed on the host: one core,
    and a pool of `cores` single-threaded processes (the reference extracts features with a process pool).  Bounded
    sample: the pool works for ~10 s."""
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    from oracle import mel_oracle as mo
    with threadpool_limits(1):                             # ONE core: no BLAS threads behind the mel matmul
        mo.mel_spectrogram(clips[0])                       # filterbank construction / FFT plan warm-up
        n1, t0 = 0, time.perf_counter()
        while n1 < 4 or time.perf_counter() - t0 < single_seconds:
            mo.mel_spectrogram(clips[n1 % len(clips)])
            n1 += 1
        dt1 = time.perf_counter() - t0
        gbs1 = sum(bytes_per_clip[i % len(clips)] for i in range(n1)) / dt1 / 1e9
        with mp.get_context('fork').Pool(cores) as pool:   # (forked inside the limit: every worker is single-threaded)
            t0 = time.perf_counter()
            pool.map(_mel_cpu_clip, clips, chunksize=1)    # workers warm + a timing of one pass
            one_pass = time.perf_counter() - t0
            reps = max(1, min(512, int(pool_seconds / max(one_pass, 1e-3))))
            work = clips * reps
            t0 = time.perf_counter()
            pool.map(_mel_cpu_clip, work, chunksize=1)
            dtp = time.perf_counter() - t0
    gbsp = sum(bytes_per_clip) * reps / dtp / 1e9
    return {'value': gbsp, 'unit': 'GB/s', 'cores': cores, 'kind': 'port',
            'clips_per_s': len(work) / dtp,
            'single_core': {'value': gbs1, 'unit': 'GB/s', 'clips_per_s': n1 / dt1},
            'sample': f'{len(work)} clip extractions ({len(clips)} distinct LJSpeech-length clips of the same workload x '
                      f'{reps}) on a pool of {cores} single-threaded processes in {dtp:.1f} s; single_core = {n1} clips '
                      f'in {dt1:.1f} s; NumPy rfft + dense 80x513 mel matmul restatement of data/audio.py:81-92 '
                      f'(librosa unavailable offline); same algorithmic bytes per clip as the GPU figure'}


def mel_bench(args):
    """BASELINE.json configs[3]: wav -> 1024-pt STFT (hop 256, periodic Hann, reflect padding) -> 80-bin Slaney mel
    -> log, data/audio.py:72-92, as ONE batched launch over 10 000 LJSpeech-length clips generated on the device.
    A step = one pass over the rank's 10 000 clips.  value = algorithmic GB/s (4 N bytes of samples in + 320 bytes per
    frame out) summed over ranks / max-over-ranks time, against the 8 TB/s HBM roof.  Replicas only:
    every rank extracts its own clips, no collective on the data path."""
    from transformertts_amd import dp
    rank, local, world = dp.init_process_group()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    result = mel_run(torch.device('cuda', local), rank, world, args.clips, args.steps, args.warmup,
                     roofline=not args.no_roofline, cpu=not args.no_cpu_baseline)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()
        torch.distributed.destroy_process_group()


def mel_run(dev, rank, world, n_clips, steps, warmup, roofline=True, cpu=True, cpu_single_s=3.0, cpu_pool_s=10.0):
    """The mel workload on one rank (see mel_bench); returns the driver-format result dict (complete on rank 0)."""
    from transformertts_amd import ops
    from transformertts_amd.data.audio import Audio
    audio = Audio(22050, 1024, 80, 256, 1024, 0, 8000, 'MelGAN', device=str(dev))
    rng = np.random.default_rng(1234 + rank)
    lens = np.clip(rng.normal(145000, 48000, n_clips), 24000, 222000).astype(np.int64)
    total = int(lens.sum())
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    wav = torch.randn(total, device=dev, generator=gen) * 0.1
    t = torch.arange(total, device=dev, dtype=torch.float32) / 22050.0
    wav += 0.3 * torch.sin(2 * np.pi * 220.0 * t) + 0.2 * torch.sin(2 * np.pi * 1760.0 * t)
    del t
    clip_off = np.zeros(n_clips + 1, dtype=np.int64)
    clip_off[1:] = np.cumsum(lens)
    frame_off = np.zeros(n_clips + 1, dtype=np.int64)
    frame_off[1:] = np.cumsum(1 + lens // 256)
    frames = int(frame_off[-1])
    coff, foff = torch.from_numpy(clip_off).to(dev), torch.from_numpy(frame_off).to(dev)
    lo, cnt, ptr, w = audio._mel
    out = torch.empty((frames, 80), dtype=torch.float32, device=dev)

    def step():
        return ops.stft_logmel(wav, coff, foff, frames, 1024, 256, audio._window, 80, lo, cnt, ptr, w, 0, 1e-5, out=out)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # (the launch goes to torch's current stream: these events bracket the kernels)
    for _ in range(steps):
        mel = step()
    ev1.record()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / steps
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert bool(torch.isfinite(mel[:4096]).all())
    byt = 4.0 * total + 4.0 * 80 * frames              # algorithmic bytes of one pass of one rank
    sec = elapsed / steps
    result = {
        'metric': 'mel feature extraction, algorithmic GB/s (wav -> 1024-pt STFT -> 80-bin log-mel)',
        'value': byt * world / sec / 1e9, 'unit': 'GB/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': 1e3 * sec, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': f'BASELINE.json configs[3]: mel microbench, {n_clips} LJSpeech-length clips per GPU '
                               f'(lengths ~ clip(N(145000, 48000^2), 24000, 222000) samples, noise + 2 sines, generated '
                               f'on the device), 22.05 kHz, n_fft 1024, hop 256, 80 mels, fmax 8000, MelGAN log '
                               f'normaliser; one batched launch per step',
                   'clips': n_clips * world, 'frames': frames * world, 'samples': total * world,
                   'algorithmic_bytes_per_step': byt * world, 'parallelism': f'replicas x{world}'},
        'clips_per_s': n_clips * world / sec, 'frames_per_s': frames * world / sec,
    }
    if rank == 0 and roofline:
        gbs = byt / kernel_ms / 1e6
        traffic = stamped_pmc(MEL_PMC_FILE, ['stft_logmel_kernel'])
        gflop = frames * 28.6e3 / 1e9
        result['roofline'] = {
            'bound': 'hbm', 'kernel': 'stft_logmel_kernel<1024>', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
            'frac': gbs / PEAK_HBM_GBS, 'traffic': traffic,
            'traffic_note': (f'HBM bytes per launch from the committed PMC passes (profiles/{MEL_PMC_FILE})' if traffic else
                             'null: no PMC file collected on this build of the library (digest mismatch or file absent)'),
            'avg_launch_ms': kernel_ms, 'algorithmic_mb_per_launch': byt / 1e6,
            'algorithmic_gflop_per_launch': gflop, 'flop_per_byte': frames * 28.6e3 / byt,
            # the kernel's FLOP/byte (~21) sits at the fp32 vector ridge (157.3 T / 8 T): the other roof, stated
            'other_roof_frac': gflop / kernel_ms / PEAK_F32_MFMA_TFLOPS,
            'other_roof': f'fp32 vector / MFMA-f32 peak {PEAK_F32_MFMA_TFLOPS} TFLOP/s',
            'note': 'HIP events on the launch stream around the timed launches; algorithmic FLOPs = 25.6 k (1024-pt rFFT) + '
                    '3 k (sparse mel) per frame (SURVEY.md 8d): ~21 FLOP/B, at the fp32 vector ridge',
        }
    if rank == 0 and world == 1 and cpu:
        n_cpu = 32
        clips = [wav[int(clip_off[i]):int(clip_off[i + 1])].cpu().numpy() for i in range(n_cpu)]
        bpc = [4.0 * int(lens[i]) + 320.0 * (1 + int(lens[i]) // 256) for i in range(n_cpu)]
        result['cpu_baseline'] = mel_cpu_baseline(clips, bpc, usable_cpus(), cpu_single_s, cpu_pool_s)
    del wav, out
    return result


def f32_train_leg(dev, cfg, shape, dropout, step_flops, steps=5, warmup=2, precision='f32'):
    """The same train step on the exact-fp32 path (precision='f32': v_mfma_f32_32x32x2_f32 everywhere - the path that
    meets north_star's 1e-4 against the fp64 oracle, tests/test_config1_parity_gpu.py) - a short timed leg."""
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch
    model = ForwardTransformer.from_config(dict(cfg, dropout_rate=dropout, predictors_dropout=dropout, device=str(dev),
                                                seed=0, precision=precision))
    model._compile(learning_rate=1e-4)
    batch = [torch.from_numpy(a).to(dev) for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234)]
    for _ in range(warmup):
        out = model.train_step(*batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model.train_step(*batch)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    loss = float(out['loss'])
    assert np.isfinite(loss)
    del model, batch, out
    torch.cuda.empty_cache()
    tfs = step_flops / ms / 1e9 if step_flops else None
    if precision == 'bf16x3':
        return {'metric': 'mel-frames/sec (train step)', 'dtype': 'bf16x3', 'value': shape['B'] * shape['Tm'] / (ms * 1e-3),
                'unit': 'mel-frames/s', 'ms_per_step': ms, 'steps': steps, 'warmup': warmup, 'loss_after': loss, 'step_tflops': tfs,
                'note': "same workload, batch and dropout as the headline; precision='bf16x3' = the exact-fp32 path with its GEMM "
                        'family on three bf16 MFMAs per product (hi / lo splits, fp32 accumulate; attention, LayerNorm, residual '
                        'stream and optimiser are the fp32 path\'s): inside the 1e-4 contract on outputs, hidden states and losses '
                        '(tests/test_config1_parity_gpu.py: test_bf16x3_*)'}
    return {'metric': 'mel-frames/sec (train step)', 'dtype': 'f32', 'value': shape['B'] * shape['Tm'] / (ms * 1e-3),
            'unit': 'mel-frames/s', 'ms_per_step': ms, 'steps': steps, 'warmup': warmup, 'loss_after': loss,
            'step_tflops': tfs, 'step_mfma_frac': (tfs / PEAK_F32_MFMA_TFLOPS if tfs else None),
            'roof': f'exact-fp32 MFMA peak {PEAK_F32_MFMA_TFLOPS} TFLOP/s',
            'note': 'same workload, batch and dropout as the headline; the path inside the 1e-4 parity contract '
                    '(loss < 1e-6, mel 1.2e-6 against the fp64 oracle at this batch)'}


def also_legs(dev, cfg, shape, dropout, step_flops):
    """Short extra legs of the default run, after its timed region, so that the driver's record carries them: the
    exact-fp32 train step, the mel workload (BASELINE configs[3]) and predict (configs[4]).  Each leg is bounded (the
    whole set ~40 s) and isolated: a failing leg reports its error and leaves the headline line intact."""
    legs = {}

    def run(name, fn):
        t0 = time.perf_counter()
        try:
            legs[name] = fn()
        except Exception as e:                                 # noqa: BLE001 - the headline must still be printed
            legs[name] = {'error': f'{type(e).__name__}: {e}'}
        legs[name]['leg_seconds'] = time.perf_counter() - t0
        torch.cuda.empty_cache()

    run('train_step_f32', lambda: f32_train_leg(dev, cfg, shape, dropout, step_flops))
    run('train_step_bf16x3', lambda: f32_train_leg(dev, cfg, shape, dropout, step_flops, precision='bf16x3'))

    def mel():
        r = mel_run(dev, 0, 1, 10000, steps=3, warmup=1, roofline=True, cpu=True, cpu_single_s=1.5, cpu_pool_s=5.0)
        keep = ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'clips_per_s', 'frames_per_s')
        out = {k: r[k] for k in keep}
        out['workload'] = 'BASELINE.json configs[3]: 10 000 LJSpeech-length clips generated on the device, one batched launch per step'
        rf = r['roofline']
        out['roofline'] = {k: rf[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms',
                                              'algorithmic_mb_per_launch', 'other_roof_frac', 'other_roof')}
        cb = r['cpu_baseline']
        out['cpu_baseline'] = {k: cb[k] for k in ('value', 'unit', 'cores', 'kind', 'clips_per_s', 'single_core', 'sample')}
        return out
    run('mel', mel)

    def predict():
        cfg_p, Tp, cases, n_par = predict_cases('bf16', 20, graphs=(True,), with_maps=False)
        head = next(c for c in cases if c['batch'] == 1)
        big = next(c for c in cases if c['batch'] == 64)
        rf = predict_roofline(cfg_p, Tp, head, big, n_par)
        return {'metric': 'predict p50 latency, 400 phonemes -> ~2 280 frames, hipGraph-captured', 'unit': 'ms',
                'workload': 'BASELINE.json configs[4]', 'dtype': 'bf16',
                'batch1': {k: head[k] for k in ('p50_ms', 'p90_ms', 'rtf_p50', 'frames', 'mel_frames_per_s')},
                'batch64': {k: big[k] for k in ('p50_ms', 'p90_ms', 'rtf_p50', 'frames', 'mel_frames_per_s')},
                'roofline': {'batch1_hbm_frac': rf['frac'], 'batch1_gbs': rf['achieved'],
                             'batch64_mfma_frac': rf['batch64']['frac'], 'batch64_tflops': rf['batch64']['achieved']},
                'cpu_baseline': predict_cpu_baseline(cfg_p, Tp, usable_cpus(), seconds=4.0)}
    run('predict', predict)

    # the two workloads that were builder-run files only until round 4, each in its OWN process (fresh allocator, fresh
    # plans; the child prints the same one-line JSON this process would): the bucketed ragged batches the reference
    # actually trains on (train_tts.py:149-160 over data/datasets.py:238-284) and the reference's shipped architecture
    # (config/training_config.yaml:104-118)
    def child(argv, keep):
        import subprocess
        env = dict(os.environ)
        env.pop('WORLD_SIZE', None), env.pop('RANK', None), env.pop('LOCAL_RANK', None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, capture_output=True, text=True, timeout=420, env=env)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f'rc {r.returncode}: {r.stderr.strip()[-400:]}')
        d = json.loads(lines[-1])
        return {k: d[k] for k in keep if k in d}

    run('lj_dist', lambda: child(['--workload', 'lj-dist', '--steps', '200', '--warmup', '20', '--dropout', str(dropout)],
                                 ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config',
                                  'distinct_batch_shapes', 'padding_fraction', 'padded_mel_frames_per_s',
                                  'host_stall_ms_per_step', 'host_issue_ms_per_step', 'allocator_growth_mb',
                                  'us_per_padded_frame', 'max_shape', 'ragged_over_max_shape_per_padded_frame')))
    run('ref_default', lambda: child(['--workload', 'ref-default', '--steps', '20', '--warmup', '4', '--no-cpu-baseline',
                                      '--no-attention-maps', '--no-also', '--dropout', str(dropout)],
                                     ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config',
                                      'host_issue_ms_per_step', 'roofline')))

    def dp1():
        # the RCCL path alive on one GPU (tools/probe_rccl_world1.py: a one-rank nccl group with the collectives forced - both
        # all-reduce buckets issued from the real hook, broadcast, joins - against the plain model, bit for bit)
        import subprocess
        env = dict(os.environ)
        env.pop('WORLD_SIZE', None), env.pop('RANK', None), env.pop('LOCAL_RANK', None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'probe_rccl_world1.py'), '--steps', '6'],
                           capture_output=True, text=True, timeout=300, env=env)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
        if not lines:
            raise RuntimeError(f'rc {r.returncode}: {r.stderr.strip()[-400:]}')
        d = json.loads(lines[-1])
        return {'what': 'one-rank RCCL group, TTSMI_DP_FORCE_COLLECTIVES=1: the two all-reduce buckets of the batch-DP step issued '
                        'from the real backward hook; bit_identical = parameters and loss equal the plain step after every step',
                **{k: d[k] for k in ('backend', 'steps', 'bit_identical', 'max_abs_param_diff', 'ms_per_step_plain',
                                     'ms_per_step_with_collectives', 'grad_bytes')}}
    run('dp1_forced_collectives', dp1)
    rd = legs.get('ref_default', {})
    if 'roofline' in rd:               # the line stays readable: the step's own roof, not the per-kernel table
        rf = rd['roofline']
        rd['roofline'] = {k: rf.get(k) for k in ('step_gflop', 'step_tflops', 'step_mfma_frac', 'kernel', 'bound', 'frac',
                                                 'main_stream_launch_ms', 'side_stream_launch_ms')}
    return legs


# reference config/training_config.yaml:22-23 (mel-length buckets and their batch sizes)
LJ_BUCKET_BOUNDARIES = [200, 300, 400, 500, 600, 700, 800, 900, 1000, 1200]
LJ_BUCKET_BATCH_SIZES = [64, 42, 32, 25, 21, 18, 16, 14, 12, 6, 1]


def lj_dist_samples(n, seed, content_seed, mel_channels=80):
    """SURVEY.md 8d 'LJ-dist' set: Tp_b ~ U{60..200}, Tm_b = sum(dur_b) with mean duration 4.5 frames per phoneme
    (jittered +-25 %), capped at 900; sample 0 at both maxima.  Lengths come from `seed` (shared by all ranks: equal
    shapes per step), contents from `content_seed`.  Returned in the trainer's component order (data/datasets.py:153-169:
    mel, phonemes, durations, pitch, name); the name carries the frame count for the host-side tally."""
    rl, rc = np.random.default_rng(seed), np.random.default_rng(content_seed)
    out = []
    for i in range(n):
        tp = 200 if i == 0 else int(rl.integers(60, 201))
        tm = 900 if i == 0 else int(min(900, max(tp, round(4.5 * tp * float(rl.uniform(0.75, 1.25))))))
        dur = rc.multinomial(tm, np.full(tp, 1.0 / tp)).astype(np.int32)
        pit = rc.standard_normal(tp).astype(np.float32)
        pit[rc.random(tp) < 0.3] = 0.0
        mel = np.clip(rc.normal(-5.0, 2.0, size=(tm, mel_channels)), -11.5129, 2.0).astype(np.float32)
        tok = rc.integers(1, 127, size=tp).astype(np.int32)
        out.append((mel, tok, dur, pit, str(tm)))
    return out


def lj_dist_bench(args):
    """`--workload lj-dist`: the ragged set of SURVEY.md 8d fed through the bucketed batch producer
    (transformertts_amd/data/datasets.py = reference data/datasets.py:238-284 with the bucket table of
    config/training_config.yaml:22-23): every step is a NEW (B, Tp_max, Tm_max) - plans re-bind, keep-bit tables are
    re-sized, the producer thread pads / pins / copies ahead.  value = REAL (unpadded) mel frames per second; beside it
    the padded rate, the host's stall in next_batch(), allocator growth, and the max-shape step of the same process for
    the ms-per-padded-frame comparison."""
    from transformertts_amd import dp
    from transformertts_amd.data.datasets import Dataset
    from transformertts_amd.model.models import ForwardTransformer
    from transformertts_amd.utils.synthetic import synthetic_batch
    rank, local, world = dp.init_process_group()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cfg, shape = workload_config('configs[1]')
    cfg = dict(cfg, dropout_rate=args.dropout, predictors_dropout=args.dropout, device=str(dev), seed=0,
               precision=args.precision)
    model = ForwardTransformer.from_config(cfg)
    model._compile(learning_rate=1e-4)
    wrapped = dp.DataParallel(model)
    samples = lj_dist_samples(args.lj_samples, 1234, 4321 + rank)
    ds = Dataset(samples=list(range(len(samples))), preprocessor=lambda i: samples[i],
                 len_function=lambda mel, *_: int(mel.shape[0]), bucket_boundaries=LJ_BUCKET_BOUNDARIES,
                 bucket_batch_sizes=LJ_BUCKET_BATCH_SIZES, shuffle=True, drop_remainder=True, seed=42, device=dev, prefetch=4)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    shapes, real, padded, stall = set(), 0, 0, 0.0
    preloaded = None
    if args.lj_preload:          # diagnosis: the same batches, produced BEFORE the timed loop (no producer thread beside it)
        preloaded = [ds.next_batch() for _ in range(args.steps + args.warmup)]
        ds.close()
        torch.cuda.synchronize()

    def one(count):
        nonlocal real, padded, stall
        t0 = time.perf_counter()
        mel, tok, dur, pit, names = preloaded.pop() if preloaded is not None else ds.next_batch()
        t1 = time.perf_counter()
        B, Tm, Tp = int(mel.shape[0]), int(mel.shape[1]), int(tok.shape[1])
        out = wrapped.train_step(tok, mel, dur, pit, global_shape=(B * world, Tp, Tm), reduce_losses=False)
        if count:
            stall += t1 - t0
            shapes.add((B, Tp, Tm))
            real += sum(int(n) for n in names)
            padded += B * Tm
        return out

    for _ in range(args.warmup):
        one(False)
    sync()
    reserved0 = torch.cuda.memory_reserved(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one(True)
    host = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    reserved1 = torch.cuda.memory_reserved(dev)
    loss = float(out['loss']) * world
    assert np.isfinite(loss), 'non-finite loss'
    ds.close()
    # the max-shape step of the SAME process (plans already at capacity): ms per padded frame to compare with
    batch = [torch.from_numpy(a).to(dev) for a in synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234 + rank)]
    gshape = (shape['B'] * world, shape['Tp'], shape['Tm'])
    n_max = 0 if args.lj_skip_max_shape else 20          # (profiling runs: only the ragged steps in the trace)
    for _ in range(3 if n_max else 0):
        wrapped.train_step(*batch, global_shape=gshape, reduce_losses=False)
    sync()
    t1 = time.perf_counter()
    for _ in range(n_max):
        wrapped.train_step(*batch, global_shape=gshape, reduce_losses=False)
    sync()
    ms_max = 1e3 * (time.perf_counter() - t1) / max(n_max, 1)
    us_per_padded = 1e6 * elapsed / padded
    us_per_padded_max = 1e3 * ms_max / (shape['B'] * shape['Tm'])
    result = {
        'metric': 'mel-frames/sec (train step), real unpadded frames, bucketed ragged batches', 'value': real * world / elapsed,
        'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': f'SURVEY.md 8d LJ-dist set: {len(samples)} samples per GPU, Tp ~ U{{60..200}}, mean duration 4.5 '
                               f'frames, mel length <= 900, through the bucketed batch producer (boundaries '
                               f'{LJ_BUCKET_BOUNDARIES}, batch sizes {LJ_BUCKET_BATCH_SIZES}: reference '
                               f'config/training_config.yaml:22-23), ForwardTransformer configs[1] architecture, dropout '
                               f'{args.dropout}', 'parallelism': f'dp{world}', 'loss_after': loss},
        'distinct_batch_shapes': len(shapes), 'real_frames': real * world, 'padded_frames': padded * world,
        'padding_fraction': 1.0 - real / padded,
        'padded_mel_frames_per_s': padded * world / elapsed,
        'host_stall_ms_per_step': 1e3 * stall / args.steps,
        'host_issue_ms_per_step': 1e3 * host / args.steps,
        'allocator_growth_mb': (reserved1 - reserved0) / 1e6, 'allocator_reserved_mb': reserved1 / 1e6,
        'batches_preloaded': bool(args.lj_preload),
        'us_per_padded_frame': us_per_padded,
        'max_shape': {'ms_per_step': ms_max, 'us_per_padded_frame': us_per_padded_max, 'steps': n_max},
        'ragged_over_max_shape_per_padded_frame': us_per_padded / us_per_padded_max,
    }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        sync()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 40 timed steps after 5 warm-up steps (~0.25 s of GPU time): the timed region starts from an idle GPU, so
    # with 10 steps the ramp of the first one and the drain of the last were ~1 % of the figure (5.31 vs 5.25 ms)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='configs[1]')
    ap.add_argument('--dropout', type=float, default=0.1)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch override (measurement only: value is then '
                                                            'NOT the configs[1] number)')
    # BASELINE.json configs[1] is quoted in bf16: bf16 GEMM/attention operands, fp32 accumulate, fp32
    # master weights / activations / optimiser.  --precision f32 runs the exact-fp32 parity path.
    ap.add_argument('--precision', default='bf16', choices=['f32', 'bf16', 'bf16x3'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-attention-maps', dest='with_attention_maps', action='store_false',
                    help='skip the extra timed leg that also materialises the 12 attention maps')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--clips', type=int, default=10000, help='--workload mel: clips per GPU (BASELINE configs[3]: 10 000)')
    ap.add_argument('--no-also', action='store_true', help='skip the extra legs of the default run (f32 step, mel, predict)')
    ap.add_argument('--lj-samples', type=int, default=4096, help='--workload lj-dist: synthetic samples per GPU')
    ap.add_argument('--lj-skip-max-shape', action='store_true', help='--workload lj-dist: no max-shape steps after the timed loop (profiling)')
    ap.add_argument('--lj-preload', action='store_true', help='--workload lj-dist: produce every batch before the timed loop '
                                                              '(diagnosis: the step without the producer thread beside it)')
    args = ap.parse_args()

    if args.workload == 'predict':
        return predict_bench(args)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` as typed: start one rank per GPU ourselves (the same launcher the driver uses)
        import socket
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                  f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
                                  '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:])

    if args.workload == 'mel':
        return mel_bench(args)
    if args.workload == 'lj-dist':
        if args.steps == 40 and args.warmup == 5:            # this workload's defaults: >= 200 steps over many shapes
            args.steps, args.warmup = 200, 20
        return lj_dist_bench(args)

    from transformertts_amd import dp
    rank, local, world = dp.init_process_group()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    local = local % max(1, torch.cuda.device_count())     # (ranks may share a GPU in functional tests)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    from transformertts_amd.model.models import ForwardTransformer
    cfg, shape = workload_config(args.workload)
    if args.batch:
        shape = dict(shape, B=args.batch)
    cfg = dict(cfg, dropout_rate=args.dropout, predictors_dropout=args.dropout, device=str(dev), seed=0,
               precision=args.precision)
    model = ForwardTransformer.from_config(cfg)
    model._compile(learning_rate=1e-4)
    wrapped = dp.DataParallel(model)

    from transformertts_amd.utils.synthetic import synthetic_batch
    tok, mel, dur, pit = synthetic_batch(shape['B'], shape['Tp'], shape['Tm'], seed=1234 + rank)
    batch = [torch.from_numpy(a).to(dev) for a in (tok, mel, dur, pit)]      # resident in HBM

    # the batch is formed globally (equal shards of one bucket): the global shape is known on the host, so no per-step
    # shape exchange; the four loss scalars are summed over ranks only for the value reported at the end
    gshape = (shape['B'] * world, shape['Tp'], shape['Tm'])

    def step(reduce_losses=False):
        return wrapped.train_step(*batch, global_shape=gshape, reduce_losses=reduce_losses)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    host_issue = time.perf_counter() - t0        # the host has ENQUEUED every step by now (no sync inside a step)
    sync()
    elapsed = time.perf_counter() - t0
    # The loop above measures the host's enqueue time INCLUDING the runtime's back-pressure: once the host is ~10 steps ahead
    # of a GPU-bound step its launches block (tools/debug/cstep_host.py: 2.5 of 2.6 ms per step inside the C calls at the
    # max shape, 1.1 ms when the GPU keeps up).  What the host NEEDS per step is a short burst into an idle queue:
    burst = 6
    t1 = time.perf_counter()
    for _ in range(burst):
        out = step()
    host_burst = time.perf_counter() - t1
    sync()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out['loss']) * world      # a rank's share of the global-batch mean (equal shards) -> the mean
    assert np.isfinite(loss), 'non-finite loss'
    frames = shape['B'] * shape['Tm'] * world
    ms = 1e3 * elapsed / args.steps
    ranks_seen = None
    if world > 1:
        # what each rank's collective library saw: backend "nccl" is RCCL on ROCm, one device per rank
        mine = {'rank': rank, 'backend': torch.distributed.get_backend(), 'world': torch.distributed.get_world_size(),
                'device': f'cuda:{local} {torch.cuda.get_device_name(local)}'}
        ranks_seen = [None] * world
        torch.distributed.all_gather_object(ranks_seen, mine)

    # the same step with the reference's full output dictionary: the 12 attention maps [B,H,T,T] of
    # model/models.py:544-549 materialised too (train_step leaves them out unless asked: reference_outputs=True)
    ms_attn = maps_alloc = None
    if args.with_attention_maps and args.precision == 'bf16':
        model.reference_outputs = True
        out_a = None
        for _ in range(3):                   # warm-up in the timed loop's own form (the previous step's output still held)
            out_a = step()
        sync()
        st0 = torch.cuda.memory_stats(dev)
        t1 = time.perf_counter()
        n_attn = max(3, args.steps // 2)
        for _ in range(n_attn):
            out_a = step()
        sync()
        ms_attn = 1e3 * (time.perf_counter() - t1) / n_attn
        st1 = torch.cuda.memory_stats(dev)
        # allocator activity INSIDE the timed maps loop (BENCH_r04: 11.3 ms at --steps 20 against 6.2 at --steps 40 -
        # 2.7 GB of fresh map tensors per step went through the allocator; now a ring of persistent buffers, models.py)
        maps_alloc = {'steps': n_attn, 'map_ring': bool(model.map_ring),
                      'reserved_growth_mb': (st1['reserved_bytes.all.current'] - st0['reserved_bytes.all.current']) / 1e6,
                      'segment_allocs': st1['segment.all.allocated'] - st0['segment.all.allocated'],
                      'alloc_retries': st1['num_alloc_retries'] - st0['num_alloc_retries'],
                      'device_mallocs': st1.get('num_device_alloc', 0) - st0.get('num_device_alloc', 0)}
        if world > 1:
            t = torch.tensor([ms_attn], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ms_attn = float(t.item())
        assert len(out_a['decoder_attention']) == len(cfg['decoder_num_heads'])
        del out_a
        model.reference_outputs = False
        model._map_bufs.clear()

    result = {
        'metric': 'mel-frames/sec (train step)', 'value': frames / (elapsed / args.steps),
        'unit': 'mel-frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': f'BASELINE.json {args.workload}: ForwardTransformer d_model=256 6+6 dense '
                               f'blocks 4 heads FFN=1024 80-mel, fwd+bwd+TF-Adam, per-GPU batch '
                               f'{shape["B"]} x {shape["Tp"]} phonemes x {shape["Tm"]} frames '
                               f'(max-shape set), dropout {args.dropout}'
                   if args.workload == 'configs[1]' else args.workload,
                   'global_batch': shape['B'] * world, 'parallelism': f'dp{world}',
                   'params': model.params.n_params, 'loss_after': loss, 'ranks_seen': ranks_seen},
        # value/ms_per_step: train_step without the 12 [B,H,T,T] attention maps in its output (they are
        # not materialised on the throughput path); the second figure is the same step returning them all
        'ms_per_step_with_attention_maps': ms_attn,
        'attention_maps_allocator': maps_alloc,
        # host time to enqueue one step (Python + ctypes launch loop); when it approaches ms_per_step the host, not
        # the GPU, bounds the step
        'host_issue_ms_per_step': 1e3 * host_issue / args.steps,
        # ... and without the runtime's back-pressure: 6 steps enqueued into an idle queue (the host's own cost per step)
        'host_issue_burst_ms_per_step': 1e3 * host_burst / burst,
        'step_issued_from': 'C++ (ttsmi_ft_train_step)' if getattr(model, '_cstep', None) is not None else 'per-layer autograd path',
    }

    if rank == 0 and not args.no_roofline:
        groups = group_records(instrumented_step(step))
        # the dominant kernel family = the one the step spends most launch time in (side-stream wgrad
        # launches overlap the main stream, so the sum of launch times exceeds the step time)
        # (the critical path is the main stream: side-stream launches are listed but not eligible)
        dom, (n, fl, by, gms) = max(((k, v) for k, v in groups.items() if k != RIDERS and not k.endswith(SIDE)),
                                    key=lambda kv: kv[1][3])
        peak_fl = peak_of(dom)
        ridge = peak_fl * 1e12 / (PEAK_HBM_GBS * 1e9)            # FLOP/byte where the two roofs meet
        hbm_bound = fl / by < ridge
        gbs, tfs = by / gms / 1e6, fl / gms / 1e9
        traffic = pmc_traffic(dom) if args.precision == 'bf16' and args.workload == 'configs[1]' else None
        result['roofline'] = {
            'bound': 'hbm' if hbm_bound else 'mfma', 'kernel': dom, 'launches_per_step': n,
            'achieved': gbs if hbm_bound else tfs, 'peak': PEAK_HBM_GBS if hbm_bound else peak_fl,
            'unit': 'GB/s' if hbm_bound else 'TFLOP/s',
            'frac': gbs / PEAK_HBM_GBS if hbm_bound else tfs / peak_fl,
            'traffic': traffic,
            'traffic_note': ('HBM bytes per launch from the committed PMC passes (profiles/' + PMC_FILE +
                             ', FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 correction applied)')
                            if traffic else 'null: no PMC file collected on this build of the library (digest mismatch or file absent)',
            'avg_launch_ms': gms / n, 'algorithmic_mb_per_launch': by / n / 1e6,
            'algorithmic_gflop_per_launch': fl / n / 1e9, 'flop_per_byte': fl / by, 'ridge_flop_per_byte': ridge,
            'other_roof_frac': tfs / peak_fl if hbm_bound else gbs / PEAK_HBM_GBS,
            'per_kernel': {k: {'launches': v[0], 'gflop': v[1] / 1e9, 'algorithmic_mb': v[2] / 1e6, 'ms': v[3],
                               'tflops': (v[1] / v[3] / 1e9 if v[1] else None),
                               'gbs': (v[2] / v[3] / 1e6 if v[2] else None)} for k, v in groups.items()},
            # launches seen by the observer hook (kernel launches issued from inside the library), NOT C-ABI calls: the
            # stack launchers issue ~90 launches per call (the step is ~107 C-ABI calls, profiles/r04_host_split.txt)
            'hooked_launches_per_step': sum(v[0] for v in groups.values()),
            'launch_ms_note': 'sums of per-launch HIP-event times from ONE extra instrumented step (every launch '
                              'bracketed by two events while the weight-gradient stream contends): slower than the '
                              'timed steps, so main_stream_launch_ms may exceed ms_per_step',
            'step_totals': pmc_step_totals() if args.precision == 'bf16' and args.workload == 'configs[1]' else None,
            'main_stream_launch_ms': sum(v[3] for k, v in groups.items() if not k.endswith(SIDE)),
            'side_stream_launch_ms': sum(v[3] for k, v in groups.items() if k.endswith(SIDE)),
        }
        # the attention families against the roof that binds them (the vector ALU: see ATTN_VALU_CYCLES_PER_SCORE_ROW)
        if args.precision == 'bf16':
            add_attention_valu_roofs(result['roofline']['per_kernel'], cfg)
        # SURVEY 8d names the MFMA roof for the STEP (a dense contraction): algorithmic matmul FLOPs of the whole step
        # (entry points' 2 per MAC, no recompute) over the timed step, against the dense bf16 / fp32 MFMA peak
        step_flops = sum(v[1] for v in groups.values())
        peak_step = PEAK_BF16_MFMA_TFLOPS if args.precision == 'bf16' else PEAK_F32_MFMA_TFLOPS
        result['roofline']['step_gflop'] = step_flops / 1e9
        result['roofline']['step_tflops'] = step_flops / ms / 1e9
        result['roofline']['step_mfma_frac'] = step_flops / ms / 1e9 / peak_step
        # `kernel` above is the dominant family of the MAIN stream (the critical path).  The family with the most GPU
        # time overall may be the weight gradients on the second stream, which run underneath it: reported as well
        gdom, (gn, gfl, gby, gms_) = max(((k, v) for k, v in groups.items() if k.replace(SIDE, '') != RIDERS),
                                         key=lambda kv: kv[1][3])
        result['roofline']['dominant_by_gpu_time'] = {
            'kernel': gdom, 'launches_per_step': gn, 'launch_ms_per_step': gms_, 'tflops': gfl / gms_ / 1e9,
            'mfma_frac': gfl / gms_ / 1e9 / peak_of(gdom), 'gbs': gby / gms_ / 1e6, 'hbm_frac': gby / gms_ / 1e6 / PEAK_HBM_GBS,
            'stream': 'weight-gradient side stream (overlaps the main stream)' if gdom.endswith(SIDE) else 'main',
            'traffic': pmc_traffic(gdom) if args.precision == 'bf16' and args.workload == 'configs[1]' else None}
    elif world > 1 and not args.no_roofline:
        step()          # keep the collective count equal on every rank
    if world > 1:
        sync()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline({k: v for k, v in cfg.items() if k not in ('device', 'seed', 'precision', 'use_graph')},
                                              shape, usable_cpus())
    # the other measurements the contract names, as short legs AFTER the timed region (driver-visible: same JSON line)
    if (rank == 0 and world == 1 and not args.no_also and not args.no_roofline and args.workload == 'configs[1]'
            and args.precision == 'bf16' and not args.batch):
        step_flops = result.get('roofline', {}).get('step_gflop', 0.0) * 1e9
        del model, wrapped, batch, out
        torch.cuda.empty_cache()
        base_cfg, _ = workload_config(args.workload)
        result['also'] = also_legs(dev, base_cfg, shape, args.dropout, step_flops)
        # LAST key of the line (the driver keeps the last 2 000 characters of stdout): one number per leg
        al = result['also']
        g = lambda leg, *path: (lambda v: round(v, 4) if isinstance(v, float) else v)(
            __import__('functools').reduce(lambda d, k: d.get(k) if isinstance(d, dict) else None, path, al.get(leg, {})))
        result['summary'] = {
            'train_step_bf16_ms': round(result['ms_per_step'], 4), 'host_issue_burst_ms': round(result['host_issue_burst_ms_per_step'], 4),
            'train_step_f32_ms': g('train_step_f32', 'ms_per_step'), 'train_step_bf16x3_ms': g('train_step_bf16x3', 'ms_per_step'), 'mel_gbs': g('mel', 'value'),
            'mel_hbm_frac': g('mel', 'roofline', 'frac'), 'predict_b1_p50_ms': g('predict', 'batch1', 'p50_ms'),
            'predict_b64_p50_ms': g('predict', 'batch64', 'p50_ms'), 'lj_dist_ms': g('lj_dist', 'ms_per_step'),
            'lj_dist_real_frames_per_s': g('lj_dist', 'value'), 'lj_dist_over_max_shape': g('lj_dist', 'ragged_over_max_shape_per_padded_frame'),
            'ref_default_ms': g('ref_default', 'ms_per_step'), 'ref_default_host_ms': g('ref_default', 'host_issue_ms_per_step'),
            'dp1_bit_identical': g('dp1_forced_collectives', 'bit_identical'),
            'dp1_ms_with_collectives': g('dp1_forced_collectives', 'ms_per_step_with_collectives'),
            'leg_errors': [k for k, v in al.items() if isinstance(v, dict) and 'error' in v]}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
