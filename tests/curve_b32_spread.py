"""How far do 300-step training trajectories at the BENCHMARKED batch (32 x 200 x 900) spread - and does the chain form of the
dense blocks change that?  (Measurement beside tests/test_training_curve_gpu.py, which asserts on ONE fp32 pair; it lives
under tests/ because it takes its config and weights from the oracle.  Run: python tests/curve_b32_spread.py)

One exact-fp32 run is the baseline.  Against it: exact fp32 from weights perturbed by one part in 10^7 (three seeds), the
bf16 path with the chain kernels (the default) and without them (chain_blocks=False), each from the unperturbed weights and
from the three perturbed sets.  Printed: the 50-step block means' relative distance to the baseline, per run.
Writes gpurun_out/r05_curve_b32_spread.json."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ft_oracle as fo                                         # noqa: E402  (weights / config of the test)
from transformertts_amd.model.models import ForwardTransformer            # noqa: E402
from transformertts_amd.utils.synthetic import learnable_batch            # noqa: E402

STEPS, BLOCK, LR, BATCH = 300, 50, 1e-4, int(os.environ.get('TTSMI_CURVE_BATCH', '32'))


def curve(precision, cfg, W, dev, **kw):
    m = ForwardTransformer.from_config(dict(cfg, precision=precision, seed=3, **kw))
    m.load_weights_dict(W)
    m._compile(learning_rate=LR)
    losses = []
    for _ in range(STEPS):
        m.set_constants(learning_rate=LR)
        losses.append(m.train_step(*dev)['loss'].clone())
    torch.cuda.synchronize()
    return np.array([float(x) for x in losses])


def main():
    cfg = dict(fo.make_config(), dropout_rate=0.0, predictors_dropout=0.0)
    W = fo.init_weights(cfg, seed=5)
    dev = [torch.from_numpy(np.asarray(a)).cuda() for a in learnable_batch(BATCH, 200, 900, seed=77)]
    sets = {'w': W}
    for s in (1, 2, 3):
        rng = np.random.default_rng(s)
        sets[f'w+1e-7 seed {s}'] = {k: (np.asarray(v) * (1.0 + 1e-7 * rng.standard_normal(np.shape(v)))).astype(np.float32)
                                    for k, v in W.items()}
    blocks = lambda c: c.reshape(STEPS // BLOCK, BLOCK).mean(axis=1)
    base = curve('f32', cfg, W, dev)
    bb = blocks(base)
    out = {'batch': BATCH, 'steps': STEPS, 'block': BLOCK, 'lr': LR, 'f32_block_means': bb.tolist(), 'runs': []}
    print('f32 baseline block means', np.round(bb, 4).tolist())

    def report(name, c):
        rel = np.abs(blocks(c) - bb) / bb
        mean_rel = abs(c[BLOCK:].mean() - base[BLOCK:].mean()) / base[BLOCK:].mean()
        out['runs'].append({'run': name, 'block_rel': rel.tolist(), 'mean_rel_50_299': float(mean_rel), 'final_block': float(blocks(c)[-1])})
        print(f'{name:42s} blocks % ' + ' '.join(f'{100 * r:5.2f}' for r in rel) + f'   mean 50..299 {100 * mean_rel:5.2f} %', flush=True)

    for name, Wx in sets.items():
        if name != 'w':
            report('f32 ' + name, curve('f32', cfg, Wx, dev))
    for name, Wx in sets.items():
        report('bf16 chains ' + name, curve('bf16', cfg, Wx, dev))
    for name, Wx in sets.items():
        report('bf16 no chains ' + name, curve('bf16', cfg, Wx, dev, chain_blocks=False))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, 'r05_curve_b32_spread.json'), 'w'))


if __name__ == '__main__':
    main()
