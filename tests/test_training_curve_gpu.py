"""-m gpu: the bf16 throughput path TRAINS like the exact-fp32 parity path over hundreds of steps, not just one.

Every other bf16 check compares a single step (or two) with the oracle.  The reference's use of the step is a 100 000-step
loop (train_tts.py:149-160: set_constants(learning_rate) -> train_step -> ...), so what matters in the end is whether the
rounding of the bf16 path changes where training goes.  Here both precisions start from the same seeded weights and take
300 Adam steps at the reference's learning rate (1e-4, config/training_config.yaml:129-131) on one fixed ragged batch of
the benchmarked architecture (BASELINE.json configs[1]: d_model 256, 6+6 dense blocks, 4 heads, FFN 1024), dropout 0 so
that the two runs see the same function: the bf16 loss must stay within 2 % of the fp32 loss at every 50th step and at
the end, and both must have learned the batch (final loss below 0.6 x the initial one).  The batch is `learnable_batch`:
durations, pitch and mel frames are functions of the token ids - the noise targets of the throughput benchmark cannot be
fitted (a first version of this test on them plateaued at 0.62 x the initial loss in both precisions, at the noise's mean
absolute deviation; so did 1e-3 on this batch: a 12-block post-LayerNorm stack without warm-up only learns the biases at
that rate.  The torch-CPU oracle at this architecture and rate goes 9.9 -> 0.85 in 300 steps).
A single step's loss wiggles by a few per cent around the trend, and the curve has spikes (Adam at a constant rate on a
post-LayerNorm stack without warm-up): two trajectories that differ by ONE rounding error separate and then wiggle and
spike independently.  So (i) each checkpoint compares the MEAN over the 10 steps around it, and (ii) the test measures
how far two exact-fp32 runs drift apart when the initial weights of one are perturbed by one part in 10^7 (the "chaos
floor" of this batch and rate) and holds bf16 to max(2 %, twice that floor): measured on the first version of this test,
bf16 and fp32 window means agreed to 0.3-1.6 % except around a loss spike near step 100 (20 %) that the two runs took at
different steps.  The raw values are printed beside the window means."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

STEPS, EVERY, WINDOW = 300, 50, 10
LR = float(os.environ.get('TTSMI_CURVE_LR', '1e-4'))          # (measurement knob: the curve at another learning rate)


def _curve(precision, cfg, W, batch):
    from transformertts_amd.model.models import ForwardTransformer
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


def _checkpoints(curve):
    """(raw loss, mean over the WINDOW steps around it) at steps 0, EVERY, 2 EVERY, .. and the last one"""
    out = []
    for k in list(range(0, STEPS, EVERY)) + [STEPS - 1]:
        lo = max(0, min(k - WINDOW // 2, STEPS - WINDOW))
        out.append((float(curve[k]), float(curve[lo:lo + WINDOW].mean())))
    return out


def test_bf16_training_curve_tracks_fp32_over_300_steps():
    from transformertts_amd.utils.synthetic import learnable_batch
    cfg = dict(fo.make_config(), dropout_rate=0.0, predictors_dropout=0.0)
    W = fo.init_weights(cfg, seed=5)
    batch = learnable_batch(8, 200, 900, seed=77)
    f32 = _curve('f32', cfg, W, batch)
    bf16 = _curve('bf16', cfg, W, batch)
    rng = np.random.default_rng(1)
    Wp = {k: (np.asarray(v) * (1.0 + 1e-7 * rng.standard_normal(np.shape(v)))).astype(np.float32) for k, v in W.items()}
    f32p = _curve('f32', cfg, Wp, batch)                        # the same precision, weights off by one part in 10^7
    cf, cb, cp = _checkpoints(f32), _checkpoints(bf16), _checkpoints(f32p)
    rel_raw = [abs(a[0] - b[0]) / b[0] for a, b in zip(cb, cf)]
    rel = [abs(a[1] - b[1]) / b[1] for a, b in zip(cb, cf)]
    floor = [abs(a[1] - b[1]) / b[1] for a, b in zip(cp, cf)]
    line = {'steps': STEPS, 'every': EVERY, 'window': WINDOW, 'lr': LR, 'f32': [c[0] for c in cf], 'bf16': [c[0] for c in cb],
            'f32_window_mean': [c[1] for c in cf], 'bf16_window_mean': [c[1] for c in cb],
            'f32_perturbed_window_mean': [c[1] for c in cp], 'rel_raw': rel_raw, 'rel': rel, 'chaos_floor': floor}
    print('\nbf16 vs f32 training curve', json.dumps(line))
    d = os.path.join(os.path.dirname(HERE), 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'bf16_vs_f32_curve.json'), 'w') as f:
            json.dump(dict(line, f32_curve=f32.tolist(), bf16_curve=bf16.tolist(), f32_perturbed_curve=f32p.tolist()), f)
    assert np.isfinite(f32).all() and np.isfinite(bf16).all()
    assert f32[-WINDOW:].mean() < 0.6 * f32[0] and bf16[-WINDOW:].mean() < 0.6 * bf16[0], (cf, cb)
    bound = max(2e-2, 2.0 * max(floor))
    assert max(rel) < bound, (rel, floor, rel_raw)
    # and over the whole run: the mean loss of the last 100 steps (spikes average out) within 3 %
    tail = abs(bf16[-100:].mean() - f32[-100:].mean()) / f32[-100:].mean()
    assert tail < max(3e-2, 2.0 * abs(f32p[-100:].mean() - f32[-100:].mean()) / f32[-100:].mean()), tail
