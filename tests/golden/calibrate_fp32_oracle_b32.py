#!/usr/bin/env python
"""What does plain fp32 arithmetic give at B = 32?  Runs the torch-CPU oracle in FLOAT32 (same eight 4-sample groups as
make_config1_b32_golden.py, group gradients averaged in fp32) and reports its error against the fp64 golden in the
vocabulary of tests/test_config1_parity_gpu.py.  This calibrates the gradient bounds of the f32 B = 32 parity test: the
encoder-side gradients sit behind the pitch predictor's last LayerNorm, whose backward cancels (csrc/common.h, wave_sum
comment), so their fp32 error grows with the number of rows summed - in ANY fp32 implementation.

    python tests/golden/calibrate_fp32_oracle_b32.py [maxshape|ragged]      # ~3 min per batch
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import ft_oracle as fo  # noqa: E402
import make_config1_b32_golden as g32  # noqa: E402
from make_config1_golden import PERTURB, WEIGHT_SEED, sample_index  # noqa: E402


def main():
    tags = sys.argv[1:] or ['maxshape', 'ragged']
    with np.load(os.path.join(HERE, 'ft_config1_b32.npz')) as z:
        gold = {k: z[k] for k in z.files}
    cfg = fo.make_config()
    W = fo.init_weights(cfg, seed=WEIGHT_SEED, perturb=PERTURB)
    real = fo.ForwardTransformerOracle

    class F32(real):
        def __init__(self, cfg, W, dtype=torch.float64, **kw):
            super().__init__(cfg, W, torch.float32, **kw)
    for tag in tags:
        batch = fo.synthetic_batch(*g32.SHAPE, **g32.BATCHES[tag])
        fo.ForwardTransformerOracle = F32
        try:
            res = g32.run_groups(cfg, W, batch)
        finally:
            fo.ForwardTransformerOracle = real
        names = list(res['grads'])
        gmax = max(float(gold[f'{tag}::gstat::{k}'][0]) for k in names)
        rep = {'batch': tag, 'loss': abs(res['loss'] - float(gold[f'{tag}::loss'])) / float(gold[f'{tag}::loss'])}
        worst, worst_vec = ('', 0.0), ('', 0.0)
        for k in names:
            a = res['grads'][k].astype(np.float64).reshape(-1)
            absmax = float(gold[f'{tag}::gstat::{k}'][0])
            e = float(np.abs(a[sample_index(k, a.size)] - gold[f'{tag}::g::{k}']).max() / max(absmax, 1e-3 * gmax))
            if k == 'embedding':
                rep['grad_embedding'] = e
            elif res['grads'][k].ndim <= 1:
                worst_vec = max(worst_vec, (k, e), key=lambda t: t[1])
            else:
                worst = max(worst, (k, e), key=lambda t: t[1])
        rep['grad_worst'], rep['grad_vec_worst'] = worst, worst_vec
        print(json.dumps(rep), flush=True)


if __name__ == '__main__':
    main()
