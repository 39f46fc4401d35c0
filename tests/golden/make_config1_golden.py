#!/usr/bin/env python
"""Frozen fp64 oracle results at the BENCHMARKED architecture (BASELINE.json configs[1]: d_model 256, 6+6
dense blocks, 4 heads, FFN 1024, predictors [256,226], reference model/models.py:464-482 train step) for
B = 4 x 200 phonemes x 900 frames - one LJ-dist ragged batch and one max-shape batch.

The fp64 oracle needs ~75 s and ~4 GB per train step at this size, so the GPU box does not run it: the
`-m gpu` tests (tests/test_config1_parity_gpu.py) rebuild the seeded weights and inputs (cheap, NumPy) and
compare the HIP path with what is frozen here:
  * forward: mel (in full, fp32), predicted duration / pitch, the loss and its three terms;
  * per-block hidden states (`taps`): SAMPLES positions per block output, fixed seeded index sets - the
    bf16 path's error is reported per layer depth from these;
  * every gradient: |g|_max, |g|_2, sum(g) and SAMPLES seeded elements of each of the 223 variables.

    python tests/golden/make_config1_golden.py        # ~3 min; rewrites tests/golden/ft_config1.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ft_oracle as fo  # noqa: E402

SAMPLES = 1024
WEIGHT_SEED, PERTURB = 2025, 0.02
BATCHES = {'ragged': dict(seed=31, ragged=True), 'maxshape': dict(seed=32, ragged=False)}
SHAPE = (4, 200, 900)


def sample_index(name: str, numel: int) -> np.ndarray:
    """The fixed element subset of tensor `name` (all of it when it has <= SAMPLES elements)."""
    if numel <= SAMPLES:
        return np.arange(numel)
    h = np.frombuffer(name.encode(), dtype=np.uint8).astype(np.uint64)
    seed = int((h * (np.arange(h.size, dtype=np.uint64) + 1)).sum() % (2 ** 31))
    return np.sort(np.random.default_rng(seed).choice(numel, size=SAMPLES, replace=False))


def build(cfg, W, batch):
    ref = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    ref.taps = []
    tr = ref.train_step(*batch, apply=False)
    out = dict(mel=tr['mel'].numpy().astype(np.float32), duration=tr['duration'].numpy(),
               pitch=tr['pitch'].numpy(), loss=np.float64(tr['loss']),
               losses=np.array([float(tr['losses'][k]) for k in ('mel', 'duration', 'pitch')]))
    for name, t in ref.taps:
        a = t.numpy().reshape(-1)
        out[f'tap::{name}'] = a[sample_index('tap::' + name, a.size)]
        out[f'tapmax::{name}'] = np.float64(np.abs(a).max())
    for k, g in tr['grads'].items():
        a = g.numpy().reshape(-1)
        out[f'g::{k}'] = a[sample_index(k, a.size)]
        out[f'gstat::{k}'] = np.array([np.abs(a).max(), np.sqrt((a * a).sum()), a.sum()])
    return out


def main():
    torch.manual_seed(0)
    cfg = fo.make_config()                      # BASELINE.json configs[1] architecture
    W = fo.init_weights(cfg, seed=WEIGHT_SEED, perturb=PERTURB)
    out = {}
    for tag, kw in BATCHES.items():
        batch = fo.synthetic_batch(*SHAPE, **kw)
        for k, v in build(cfg, W, batch).items():
            out[f'{tag}::{k}'] = v
        print(tag, 'loss', float(out[f'{tag}::loss']), flush=True)
    path = os.path.join(HERE, 'ft_config1.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
