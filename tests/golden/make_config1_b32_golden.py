#!/usr/bin/env python
"""Frozen fp64 oracle results for ONE TRAIN STEP AT THE BENCHMARKED SIZE: BASELINE.json configs[1], batch 32 x 200
phonemes x 900 frames (reference model/models.py:464-482) - the launch shapes bench.py times (M_dec = 28 800 rows,
M_enc = 6 400), which select kernel variants that the B = 4 golden (make_config1_golden.py) never reaches.

A B = 32 fp64 step would need ~32 GB; samples of a batch are independent in this model (no cross-sample op but the
loss mean and the zero padding to `max_b sum(dur)`), so the golden is assembled from EIGHT fp64 runs of four samples:
  * every group is run on arrays padded to the GLOBAL shapes and its expanded sequence is zero-padded to the global
    `max_b sum(dur)` (what `Expand` produces for the whole batch, model/layers.py:557-565) - padded query rows do
    attend, so the padding has to be there for the group's outputs to equal the batch's;
  * outputs are concatenated on the batch axis; each loss is a mean over equal-sized groups, so the batch loss is the
    mean of the group losses and the batch gradient the mean of the group gradients.
tests/test_oracle.py::test_group_assembly_equals_the_whole_batch checks exactly this assembly against one whole-batch
oracle run at a size the CPU suite affords.

    python tests/golden/make_config1_b32_golden.py [--jobs 2]     # ~12 min; rewrites tests/golden/ft_config1_b32.npz
"""
import argparse
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

from oracle import ft_oracle as fo  # noqa: E402
from make_config1_golden import PERTURB, SAMPLES, WEIGHT_SEED, sample_index  # noqa: E402

SHAPE = (32, 200, 900)
GROUP = 4
BATCHES = {'maxshape': dict(seed=42, ragged=False), 'ragged': dict(seed=41, ragged=True)}
MEL_SAMPLES = 65536


def mel_sample_index(numel: int) -> np.ndarray:
    return np.sort(np.random.default_rng(77).choice(numel, size=min(MEL_SAMPLES, numel), replace=False))


def run_groups(cfg, W, batch, group=GROUP, collect=None):
    """The assembly described in the module docstring.  Returns dict(mel, duration, pitch [concatenated, fp64],
    loss, losses[3], grads{name: fp64 mean over groups}, taps{name: concatenated}).  `collect(tag, arr)` may be given
    to subsample big tensors group by group instead of keeping them."""
    tokens, mel, durs, pitch = batch
    B = tokens.shape[0]
    assert B % group == 0
    out_len = int(fo.expand_indices_np(durs[..., None])[2])
    real_expand = fo.expand_torch

    def expand_padded(x, dimensions):
        y = real_expand(x, dimensions)
        return torch.nn.functional.pad(y, (0, 0, 0, out_len - y.shape[1]))

    acc = dict(mel=[], duration=[], pitch=[], loss=[], losses=[], taps={}, grads=None)
    fo.expand_torch = expand_padded
    try:
        for g0 in range(0, B, group):
            sl = slice(g0, g0 + group)
            ref = fo.ForwardTransformerOracle(cfg, W, torch.float64)
            ref.taps = []
            tr = ref.train_step(tokens[sl], mel[sl], durs[sl], pitch[sl], apply=False)
            acc['mel'].append(tr['mel'].numpy())
            acc['duration'].append(tr['duration'].numpy())
            acc['pitch'].append(tr['pitch'].numpy())
            acc['loss'].append(float(tr['loss']))
            acc['losses'].append([float(tr['losses'][k]) for k in ('mel', 'duration', 'pitch')])
            for name, t in ref.taps:
                acc['taps'].setdefault(name, []).append(t.numpy())
            if acc['grads'] is None:
                acc['grads'] = {k: v.numpy().copy() for k, v in tr['grads'].items()}
            else:
                for k, v in tr['grads'].items():
                    acc['grads'][k] += v.numpy()
    finally:
        fo.expand_torch = real_expand
    n = B // group
    return dict(mel=np.concatenate(acc['mel']), duration=np.concatenate(acc['duration']),
                pitch=np.concatenate(acc['pitch']), loss=float(np.mean(acc['loss'])),
                losses=np.mean(np.asarray(acc['losses']), axis=0),
                grads={k: v / n for k, v in acc['grads'].items()},
                taps={k: np.concatenate(v) for k, v in acc['taps'].items()})


def freeze(res) -> dict:
    out = dict(duration=res['duration'], pitch=res['pitch'], loss=np.float64(res['loss']), losses=res['losses'])
    m = res['mel'].reshape(-1)
    out['mel_samples'] = m[mel_sample_index(m.size)]
    out['mel_stat'] = np.array([np.abs(m).max(), np.sqrt((m * m).sum()), m.sum()])
    for name, t in res['taps'].items():
        a = t.reshape(-1)
        out[f'tap::{name}'] = a[sample_index('tap::' + name, a.size)]
        out[f'tapmax::{name}'] = np.float64(np.abs(a).max())
    for k, g in res['grads'].items():
        a = g.reshape(-1)
        out[f'g::{k}'] = a[sample_index(k, a.size)]
        out[f'gstat::{k}'] = np.array([np.abs(a).max(), np.sqrt((a * a).sum()), a.sum()])
    return out


def _one(tag):
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    cfg = fo.make_config()
    W = fo.init_weights(cfg, seed=WEIGHT_SEED, perturb=PERTURB)
    batch = fo.synthetic_batch(*SHAPE, **BATCHES[tag])
    res = freeze(run_groups(cfg, W, batch))
    print(tag, 'loss', float(res['loss']), flush=True)
    return tag, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--jobs', type=int, default=2)
    a = ap.parse_args()
    out = {}
    with ProcessPoolExecutor(max_workers=a.jobs) as ex:
        for tag, res in ex.map(_one, list(BATCHES)):
            for k, v in res.items():
                out[f'{tag}::{k}'] = v
    path = os.path.join(HERE, 'ft_config1_b32.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
