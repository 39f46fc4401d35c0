"""-m gpu parity at the BENCHMARKED architecture (BASELINE.json configs[1]: d_model 256, 6+6 dense blocks, 4 heads,
FFN 1024, predictors [256,226]) - the configuration bench.py times - against the fp64 oracle's frozen results
(tests/golden/ft_config1.npz, written by tests/golden/make_config1_golden.py; the fp64 run takes ~75 s and 4 GB per
step, so it is frozen rather than repeated here).  Reference: model/models.py:464-482 (_train_step).

Both precisions, one LJ-dist ragged batch and one max-shape batch of 4 x 200 phonemes x 900 frames:
  * precision='f32' (the parity path): loss / losses / mel / duration / pitch within 1e-4 relative, every one of
    the 223 gradients within 2e-4 of its own scale (sampled elements + L2 norm + sum of each tensor);
  * precision='bf16' (the path bench.py measures): the same quantities against the SAME oracle with the bf16
    bounds below, and the error of every block's output reported per layer depth (12 stacked bf16 blocks on an
    fp32 residual stream is where rounding accumulates)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_config1_golden as g1  # noqa: E402

pytestmark = pytest.mark.gpu

# precision -> bounds.  'fwd': mel / duration / pitch (max abs error over max abs value); 'loss': the total and the
# three terms (relative); 'grad': per tensor, max abs error of the sampled elements over max(|g|_max, 1e-3 * the
# largest |g|_max of the model); 'gnorm': relative error of each tensor's L2 norm (tensors above the same floor);
# 'tap': per block output, max abs error of the samples over the tensor's max abs value.
BOUNDS = {
    # grad_vec: bias / LayerNorm vectors are column sums over all B*T rows with heavy cancellation (the FFN hidden bias
    # worst of all): fp32 summation itself is at 2-3e-4 of the tensor's scale there - the torch-CPU fp32 run of the
    # ORACLE differs from its own fp64 run by 2.7e-4 on dec.blk2.ffn.b1 - so vectors get 4e-4, matrices keep 2e-4
    'f32': dict(fwd=1e-4, loss=1e-4, grad=2e-4, grad_embedding=2e-4, grad_vec=4e-4, gnorm=2e-4, tap=1e-4),
    # bf16 bounds = about twice what the path achieves on these two batches (round 2, fused-LayerNorm build: mel /
    # duration / pitch 0.9-1.0 %, loss 2e-4, block outputs 0.2 % at depth 1 growing to 0.7 % at depth 12, matrices
    # <= 3.7 % (embedding), vectors <= 5.4 % (the positional-encoding scalars: one number, a sum with cancellation);
    # on the max-shape batch: duration 1.6 %, pitch 1.3 %, total loss 5.3e-4, the small pitch-loss component 2.0e-3)
    # The embedding table's gradient is judged on its own: it sits below all 6 encoder blocks AND is a scatter-sum over the
    # ~6 occurrences of each token, so it carries the most bf16 noise of any tensor - 3.1 % / 6.6 % (max element error
    # over max element) on the two batches, moving by a factor of two when a kernel's rounding order changes; every
    # other matrix stays below 3 %.
    'bf16': dict(fwd=3e-2, loss=4e-3, grad=6e-2, grad_embedding=1.3e-1, grad_vec=1e-1, gnorm=1e-1, tap=1.5e-2),
}
# The bf16 path AT THE BENCHMARKED BATCH (B = 32) is held to at most twice what it measures there (round-3 final build,
# gpurun_out/config1_parity.jsonl, maxshape / ragged): outputs 1.0 / 1.0 % (mel), 2.1 / 1.6 % (duration), 1.5 / 1.6 %
# (pitch); total loss 5.6e-4, pitch-loss term 2.2e-3; block outputs <= 0.67 / 0.70 %; worst weight matrix 0.93 / 1.2 %
# (enc.blk0.wq), embedding 2.2 / 2.9 %, worst vector 2.5 / 2.9 % (enc.ln.gamma), worst L2 norm 1.4 / 0.5 %.  Gradients
# at B = 32 average 8 x more rows than the B = 4 cases above, whose wider bounds (3.8 % / 6.9 % / 7.2 % measured) stay.
BOUNDS_BF16_B32 = dict(fwd=3e-2, loss=4e-3, grad=2.4e-2, grad_embedding=5.8e-2, grad_vec=5.7e-2, gnorm=2.9e-2, tap=1.4e-2)


@pytest.fixture(scope='module')
def gold():
    with np.load(os.path.join(HERE, 'golden', 'ft_config1.npz')) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='module')
def setup():
    cfg = fo.make_config()
    return cfg, fo.init_weights(cfg, seed=g1.WEIGHT_SEED, perturb=g1.PERTURB)


def _relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _run(cfg, W, tag, precision):
    from transformertts_amd.model.models import ForwardTransformer
    batch = fo.synthetic_batch(*g1.SHAPE, **g1.BATCHES[tag])
    m = ForwardTransformer.from_config(dict(cfg, precision=precision))
    m.load_weights_dict(W)
    m._compile(learning_rate=1e-3)
    m._taps = []
    out = m.train_step(*batch)
    torch.cuda.synchronize()
    return m, out


def _compare(gold, tag, m, out, bounds):
    g = lambda k: gold[f'{tag}::{k}']
    report = {}
    # forward
    report['mel'] = _relmax(out['mel'].float().cpu().numpy(), g('mel'))
    report['duration'] = _relmax(out['duration'].float().cpu().numpy(), g('duration'))
    report['pitch'] = _relmax(out['pitch'].float().cpu().numpy(), g('pitch'))
    report['loss'] = abs(float(out['loss']) - float(g('loss'))) / float(g('loss'))
    for i, k in enumerate(('mel', 'duration', 'pitch')):
        report[f'loss_{k}'] = abs(float(out['losses'][k]) - g('losses')[i]) / g('losses')[i]
    # per-block hidden states, by depth
    taps = {}
    for name, t in m._taps:
        a = t.float().cpu().numpy().reshape(-1)
        want = g(f'tap::{name}')
        taps[name] = float(np.abs(a[g1.sample_index('tap::' + name, a.size)] - want).max() / float(g(f'tapmax::{name}')))
    report['taps'] = taps
    # gradients
    grads = m.grads_dict()
    names = [k[len(tag) + 5:] for k in gold if k.startswith(f'{tag}::g::')]
    assert sorted(names) == sorted(grads), 'gradient set differs from the oracle variable set'
    gmax = max(float(g(f'gstat::{k}')[0]) for k in names)
    worst, worst_vec, worst_norm = ('', 0.0), ('', 0.0), ('', 0.0)
    for k in names:
        a = grads[k].astype(np.float64).reshape(-1)
        absmax, l2, total = g(f'gstat::{k}')
        scale = max(absmax, 1e-3 * gmax)
        e = float(np.abs(a[g1.sample_index(k, a.size)] - g(f'g::{k}')).max() / scale)
        if k == 'embedding':
            report['grad_embedding'] = e
        elif grads[k].ndim <= 1:
            if e > worst_vec[1]:
                worst_vec = (k, e)
        elif e > worst[1]:
            worst = (k, e)
        if absmax > 1e-3 * gmax:
            en = abs(float(np.sqrt((a * a).sum())) - l2) / l2
            if en > worst_norm[1]:
                worst_norm = (k, en)
    report['grad_worst'], report['grad_vec_worst'], report['gnorm_worst'] = worst, worst_vec, worst_norm
    return report


def _check(report, b):
    for k in ('mel', 'duration', 'pitch'):
        assert report[k] < b['fwd'], (k, report[k])
    for k in ('loss', 'loss_mel', 'loss_duration', 'loss_pitch'):
        assert report[k] < b['loss'], (k, report[k])
    for name, e in report['taps'].items():
        assert e < b['tap'], (name, e)
    assert report['grad_worst'][1] < b['grad'], report['grad_worst']
    assert report['grad_embedding'] < b['grad_embedding'], report['grad_embedding']
    assert report['grad_vec_worst'][1] < b['grad_vec'], report['grad_vec_worst']
    assert report['gnorm_worst'][1] < b['gnorm'], report['gnorm_worst']


def _dump(tag, precision, report):
    """Print the achieved errors (and keep them under gpurun_out/ when that scratch directory exists, so the
    numbers quoted in DESIGN.md can be copied from a round's own run)."""
    line = {'batch': tag, 'precision': precision, **{k: v for k, v in report.items()}}
    print('\nconfig1 parity', json.dumps(line))
    d = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'config1_parity.jsonl'), 'a') as f:
            f.write(json.dumps(line) + '\n')


@pytest.mark.parametrize('tag', ['ragged', 'maxshape'])
def test_f32_path_matches_the_fp64_oracle_at_the_benchmarked_architecture(gold, setup, tag):
    cfg, W = setup
    m, out = _run(cfg, W, tag, 'f32')
    report = _compare(gold, tag, m, out, BOUNDS['f32'])
    _dump(tag, 'f32', report)
    _check(report, BOUNDS['f32'])


@pytest.mark.parametrize('tag', ['ragged', 'maxshape'])
def test_bf16_path_tracks_the_fp64_oracle_at_the_benchmarked_architecture(gold, setup, tag):
    cfg, W = setup
    m, out = _run(cfg, W, tag, 'bf16')
    report = _compare(gold, tag, m, out, BOUNDS['bf16'])
    _dump(tag, 'bf16', report)
    _check(report, BOUNDS['bf16'])
    # depth profile: the error of the decoder's last block must not have exploded relative to its first
    dec = [report['taps'][f'dec.blk{i}'] for i in range(6)]
    assert dec[-1] < 20 * max(dec[0], 1e-3), dec


# ---------------------------------------------------------------------------------------------------------------------
# The same comparison AT THE BENCHMARKED BATCH (B = 32: M_dec = 28 800 rows, M_enc = 6 400) - the launch shapes bench.py
# times, which route to kernel variants the B = 4 cases above never reach (128-row full-row GEMM+LN, persistent LDS-DMA
# GEMM, 256-column K = 256 kernel, ...).  Golden: tests/golden/ft_config1_b32.npz, assembled from eight 4-sample fp64
# oracle runs by tests/golden/make_config1_b32_golden.py (assembly checked in tests/test_oracle.py).
# ---------------------------------------------------------------------------------------------------------------------
import make_config1_b32_golden as g32  # noqa: E402


@pytest.fixture(scope='module')
def gold32():
    with np.load(os.path.join(HERE, 'golden', 'ft_config1_b32.npz')) as z:
        return {k: z[k] for k in z.files}



def _run32(cfg, W, tag, precision):
    from transformertts_amd.model.models import ForwardTransformer
    batch = fo.synthetic_batch(*g32.SHAPE, **g32.BATCHES[tag])
    m = ForwardTransformer.from_config(dict(cfg, precision=precision))
    m.load_weights_dict(W)
    m._compile(learning_rate=1e-3)
    m._taps = []
    out = m.train_step(*batch)
    torch.cuda.synchronize()
    return m, out


def _compare32(gold, tag, m, out, bounds):
    mel = out['mel'].float().cpu().numpy().astype(np.float64).reshape(-1)
    idx = g32.mel_sample_index(mel.size)
    want = gold[f'{tag}::mel_samples']
    absmax, l2, total = gold[f'{tag}::mel_stat']
    mel_err = float(np.abs(mel[idx] - want).max() / absmax)
    mel_norm_err = abs(float(np.sqrt((mel * mel).sum())) - l2) / l2
    # reuse the B = 4 comparison for everything else: give it a mel that compares equal, then overwrite the entry
    shim = dict(gold)
    shim[f'{tag}::mel'] = out['mel'].float().cpu().numpy()
    report = _compare(shim, tag, m, out, bounds)
    report['mel'] = max(mel_err, mel_norm_err)
    return report


@pytest.mark.parametrize('tag', ['maxshape', 'ragged'])
def test_bf16_path_at_the_benchmarked_batch_of_32(gold32, setup, tag):
    """BASELINE configs[1] exactly as bench.py runs it (B 32 x 200 x 900, bf16 path) against the fp64 oracle; the test
    also asserts that the step went through the benchmark's kernel variants."""
    from transformertts_amd import _lib
    cfg, W = setup
    m, out = _run32(cfg, W, tag, 'bf16')
    report = _compare32(gold32, tag, m, out, BOUNDS_BF16_B32)
    _dump(tag + '_b32', 'bf16', report)
    _check(report, BOUNDS_BF16_B32)
    dec = [report['taps'][f'dec.blk{i}'] for i in range(6)]
    assert dec[-1] < 20 * max(dec[0], 1e-3), dec
    # the routers are deterministic in the shapes: the decoder-size entry points select the benchmark's variants
    l = _lib.lib()
    assert int(l.ttsmi_hgemm_ln_bwd_nparts(32 * 900)) == 225            # 128-row full-row GEMM + LayerNorm kernels
    assert bool(l._cdll.ttsmi_hgemm_k256_eligible(32 * 900, 1024, 256))


# Gradient bounds of the exact-fp32 path at B = 32.  Outputs, hidden states and losses keep the 1e-4 contract (measured
# 1e-6 / 4e-8).  The GRADIENTS are sums over 8 x more rows than at B = 4 and the encoder side sits behind the pitch
# predictor's last LayerNorm, whose backward cancels (csrc/common.h): the fp32 rounding of those sums is a property of
# fp32, not of this implementation - the torch-CPU oracle run in FLOAT32 on the same batches
# (tests/golden/calibrate_fp32_oracle_b32.py) is off the fp64 golden by embedding 3.2e-4 / 1.28e-3, matrices 1.1e-4 /
# 2.2e-4, vectors 2.6e-4 / 3.6e-4 (maxshape / ragged); the HIP path measures 7.8e-4 / 1.29e-3, 2.1e-4 / 2.1e-4,
# 3.3e-4 / 5.7e-4.  Bounds = about twice the larger of the two.
BOUNDS_F32_B32 = dict(BOUNDS['f32'], grad=5e-4, grad_embedding=3e-3, grad_vec=1.2e-3, gnorm=4e-4)


@pytest.mark.parametrize('tag', ['maxshape', 'ragged'])
def test_f32_path_at_the_benchmarked_batch_of_32(gold32, setup, tag):
    cfg, W = setup
    m, out = _run32(cfg, W, tag, 'f32')
    report = _compare32(gold32, tag, m, out, BOUNDS_F32_B32)
    _dump(tag + '_b32', 'f32', report)
    _check(report, BOUNDS_F32_B32)


# ---------------------------------------------------------------------------------------------------------------------
# precision='bf16x3' (round 6): the exact-fp32 path with its GEMM family (Dense / Conv1D forward, input and weight gradients) on
# THREE bf16 MFMAs per product (hi / lo splits, ~2^-16 per product); attention, LayerNorm, residual stream and optimiser are the
# fp32 path's.  Held to the fp32 path's own contract - outputs, hidden states and losses within 1e-4 of the fp64 oracle - at
# B = 4 and at the benchmarked B = 32; gradients to the fp32 bounds widened by the products' own error (a gradient element is a
# sum of ~10^4 products with cancellation).
# ---------------------------------------------------------------------------------------------------------------------
# measured (B = 4, ragged / maxshape): mel 1.3e-5, duration 3.5e-5, pitch 2.6e-5, loss 1.8e-6, block outputs <= 1.1e-5; worst
# weight matrix 1.5e-3 / 2.7e-3, embedding 3.9e-3 / 1.4e-3, worst vector 2.2e-3 / 1.7e-3, worst L2 norm 1.3e-3 / 1.7e-4 (the bf16
# path: 3.8e-2 / 6.9e-2 / 7.2e-2).  At B = 32: matrices 9.5e-4 / 5.4e-4, embedding 1.2e-3 / 1.3e-3, vectors 9.5e-4 / 7.3e-4.
BOUNDS_X3 = dict(BOUNDS['f32'], grad=5.5e-3, grad_embedding=8e-3, grad_vec=4.5e-3, gnorm=2.6e-3)
BOUNDS_X3_B32 = dict(BOUNDS_F32_B32, grad=2e-3, grad_embedding=4e-3, grad_vec=2.4e-3, gnorm=8e-4)


@pytest.mark.parametrize('tag', ['ragged', 'maxshape'])
def test_bf16x3_path_meets_the_fp32_contract(gold, setup, tag):
    from transformertts_amd import _lib
    cfg, W = setup
    m, out = _run(cfg, W, tag, 'bf16x3')
    assert _lib.lib().ttsmi_last_kernel() is not None
    report = _compare(gold, tag, m, out, BOUNDS_X3)
    _dump(tag, 'bf16x3', report)
    _check(report, BOUNDS_X3)


@pytest.mark.parametrize('tag', ['maxshape', 'ragged'])
def test_bf16x3_path_at_the_benchmarked_batch_of_32(gold32, setup, tag):
    cfg, W = setup
    m, out = _run32(cfg, W, tag, 'bf16x3')
    report = _compare32(gold32, tag, m, out, BOUNDS_X3_B32)
    _dump(tag + '_b32', 'bf16x3', report)
    _check(report, BOUNDS_X3_B32)
