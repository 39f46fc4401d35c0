"""-m gpu: the bf16 throughput path TRAINS like the exact-fp32 parity path over hundreds of steps, not just one.

Every other bf16 check compares a single step (or two) with the oracle.  The reference's use of the step is a 100 000-step
loop (train_tts.py:149-160: set_constants(learning_rate) -> train_step -> ...), so what matters in the end is whether the
rounding of the bf16 path changes where training goes.  Here both precisions start from the same seeded weights and take
300 Adam steps at the reference's learning rate (1e-4, config/training_config.yaml:129-131) on one fixed ragged batch of
the benchmarked architecture (BASELINE.json configs[1]: d_model 256, 6+6 dense blocks, 4 heads, FFN 1024), dropout 0 so
that the runs see the same function.  The batch is `learnable_batch`: durations, pitch and mel frames are functions of the
token ids - the noise targets of the throughput benchmark cannot be fitted (a first version of this test on them
plateaued at 0.62 x the initial loss in both precisions, at the noise's mean absolute deviation; so did 1e-3 on this
batch: a 12-block post-LayerNorm stack without warm-up only learns the biases at that rate).

What can be asked of two such trajectories.  The loss falls 10.4 -> 1.0, with spikes (Adam at a constant rate, no
warm-up): once two runs differ by ONE rounding error they take their spikes at different steps.  The test therefore
measures that sensitivity itself - a THIRD run, exact fp32 again, from weights perturbed by one part in 10^7 - and
compares means over blocks of 50 steps.  Measured (round 4): fp32 vs perturbed fp32 differ by 1.0 / 3.4 / 1.7 / 4.6 /
3.1 % in the five blocks after the first, bf16 vs fp32 by 4.1 / 6.8 / 1.9 / 6.5 / 3.3 %; a pointwise "within 2 % at every
50th step" holds for neither pair (10-step window means: fp32 pair up to 3.5 %, bf16 up to 20 % at a spike near step
100), while the first block - before the runs have separated - agrees to 0.5 %, the mean over steps 50..300 to 1.4 %
(fp32 pair 0.4 %) and the last 100 steps to 5.0 % (fp32 pair 4.0 %).  Asserted:
  * every run ends below 0.6 x its initial loss (they end near 0.1 x);
  * block 0 (steps 0..49) within 2 %;
  * every later 50-step block within max(2 %, twice the worst fp32-vs-perturbed-fp32 block);
  * the mean loss over steps 50..299 within 3 %.

Round 5: the floor is taken over THREE perturbed fp32 runs (seeds 1..3), not one.  One pair is one draw from a heavy-tailed
spread: at the benchmarked batch of 32 (tests/curve_b32_spread.py, profiles/r05_curve_b32_spread.txt) the three fp32 pairs'
worst blocks are 3.6 / 6.2 / 3.2 %, four bf16 runs with the chain kernels 8.3 / 10.9 / 5.8 / 3.1 %, four without them
3.4 / 6.3 / 3.3 / 2.7 % - and the mean over steps 50..299 is 0.2-0.9 % / 0.5-0.7 % / 0.3-1.1 %: which block a run takes its
spike in is chance, where training goes is not."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

STEPS, BLOCK = 300, 50
LR = float(os.environ.get('TTSMI_CURVE_LR', '1e-4'))          # (measurement knob: the curve at another learning rate)
BATCH = int(os.environ.get('TTSMI_CURVE_BATCH', '8'))         # (measurement knob: 32 = the benchmarked batch, profiles/r05_bf16_vs_f32_curve_b32.json)


def _curve(precision, cfg, W, batch, chain_min_rows=None):
    from transformertts_amd import ops
    from transformertts_amd.model.models import ForwardTransformer
    if chain_min_rows is not None:                             # (the chain kernels at every size, not only from 16 384 rows on)
        saved, ops.CHAIN_MIN_ROWS = ops.CHAIN_MIN_ROWS, chain_min_rows
        try:
            return _curve(precision, cfg, W, batch)
        finally:
            ops.CHAIN_MIN_ROWS = saved
    m = ForwardTransformer.from_config(dict(cfg, precision=precision, seed=3))
    m.load_weights_dict(W)
    m._compile(learning_rate=LR)
    dev = [torch.from_numpy(np.asarray(a)).cuda() for a in batch]
    losses = []
    for _ in range(STEPS):
        m.set_constants(learning_rate=LR)                      # train_tts.py:152-153 sets it every step
        losses.append(m.train_step(*dev)['loss'].clone())      # read after the loop: no host sync inside it
    torch.cuda.synchronize()
    assert m.step == STEPS
    return np.array([float(x) for x in losses])


def test_bf16_training_curve_tracks_fp32_over_300_steps():
    from transformertts_amd.utils.synthetic import learnable_batch
    cfg = dict(fo.make_config(), dropout_rate=0.0, predictors_dropout=0.0)
    W = fo.init_weights(cfg, seed=5)
    batch = learnable_batch(BATCH, 200, 900, seed=77)
    f32 = _curve('f32', cfg, W, batch)
    bf16 = _curve('bf16', cfg, W, batch)
    # the same with every dense block on the chain kernels (at the default batch of 8 no block reaches their row threshold)
    bf16c = _curve('bf16', cfg, W, batch, chain_min_rows=0)
    perturbed = []
    for s in (1, 2, 3):                                         # the same precision, weights off by one part in 10^7
        rng = np.random.default_rng(s)
        Wp = {k: (np.asarray(v) * (1.0 + 1e-7 * rng.standard_normal(np.shape(v)))).astype(np.float32) for k, v in W.items()}
        perturbed.append(_curve('f32', cfg, Wp, batch))
    f32p = perturbed[0]
    blocks = lambda c: c.reshape(STEPS // BLOCK, BLOCK).mean(axis=1)
    bf, bb, bp = blocks(f32), blocks(bf16), blocks(f32p)
    rel = np.abs(bb - bf) / bf
    relc = np.abs(blocks(bf16c) - bf) / bf
    floors = np.stack([np.abs(blocks(c) - bf) / bf for c in perturbed])
    floor = floors.max(axis=0)
    mean_rel = abs(bf16[BLOCK:].mean() - f32[BLOCK:].mean()) / f32[BLOCK:].mean()
    mean_relc = abs(bf16c[BLOCK:].mean() - f32[BLOCK:].mean()) / f32[BLOCK:].mean()
    mean_floor = abs(f32p[BLOCK:].mean() - f32[BLOCK:].mean()) / f32[BLOCK:].mean()
    line = {'steps': STEPS, 'block': BLOCK, 'lr': LR, 'batch': BATCH, 'f32_block_means': bf.tolist(), 'bf16_block_means': bb.tolist(),
            'f32_perturbed_block_means': bp.tolist(), 'bf16_vs_f32': rel.tolist(), 'f32_perturbed_vs_f32': floor.tolist(),
            'f32_perturbed_vs_f32_by_seed': floors.tolist(),
            'bf16_chains_everywhere_vs_f32': relc.tolist(), 'bf16_chains_everywhere_mean_50_299_vs_f32': float(mean_relc),
            'mean_loss_steps_50_299': {'f32': float(f32[BLOCK:].mean()), 'bf16': float(bf16[BLOCK:].mean()),
                                       'f32_perturbed': float(f32p[BLOCK:].mean()), 'bf16_vs_f32': mean_rel,
                                       'f32_perturbed_vs_f32': mean_floor},
            'every_50th_step': {'f32': f32[::BLOCK].tolist() + [float(f32[-1])], 'bf16': bf16[::BLOCK].tolist() + [float(bf16[-1])]}}
    print('\nbf16 vs f32 training curve', json.dumps(line))
    d = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'bf16_vs_f32_curve.json' if BATCH == 8 else f'bf16_vs_f32_curve_b{BATCH}.json'), 'w') as f:
            json.dump(dict(line, f32_curve=f32.tolist(), bf16_curve=bf16.tolist(), f32_perturbed_curve=f32p.tolist()), f)
    for c in [f32, bf16, bf16c] + perturbed:
        assert np.isfinite(c).all()
        assert c[-BLOCK:].mean() < 0.6 * c[0], (c[0], c[-BLOCK:].mean())
    assert rel[0] < 2e-2, rel
    assert rel[1:].max() < max(2e-2, 2.0 * floor[1:].max()), (rel, floor)
    assert mean_rel < 3e-2, (mean_rel, mean_floor)
    assert relc[0] < 2e-2, relc
    assert relc[1:].max() < max(2e-2, 2.0 * floor[1:].max()), (relc, floor)
    assert mean_relc < 3e-2, (mean_relc, mean_floor)
