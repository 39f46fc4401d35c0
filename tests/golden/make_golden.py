#!/usr/bin/env python
"""Generates the frozen known-answer fixtures under tests/golden/ (SURVEY.md section 8c.2).

The reference cannot be imported here (TensorFlow / librosa are not installable offline), so these
are SELF-CONSISTENCY vectors of the fp64 CPU oracle (oracle/ft_oracle.py, oracle/mel_oracle.py),
frozen so that (a) a later edit of the oracle that changes its numbers is caught by
tests/test_golden.py, and (b) the GPU path is compared on the GPU box against values that were
computed once, here, independent of the oracle code that travels with the repo.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

Contents
  ft_tiny.npz      BASELINE.json configs[0] shape: seeded weights (perturbed biases/LN params),
                   a ragged synthetic batch, fp64 forward outputs, the three losses, per-tensor
                   gradient max-abs/sum, and the weights after one TF-Adam step (lr 1e-3; four tensors in full + sums of all).
  lenreg.npz       duration tables incl. .5 ties (2.5 -> 2, 3.5 -> 4), zeros, an all-zero sample, and
                   the expected int32 index tables / lengths (bit-exact contract).
  mel.npz          two synthetic clips (one with len % hop == 0) and their float32 log-mel.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ft_oracle as fo  # noqa: E402
from oracle import mel_oracle as mo  # noqa: E402


def main():
    torch.manual_seed(0)
    # ---- ForwardTransformer tiny ---------------------------------------------------------------
    cfg = fo.tiny_config()
    W = fo.init_weights(cfg, seed=2024, perturb=0.02)
    batch = fo.synthetic_batch(4, 50, 200, seed=77, ragged=True)
    ref = fo.ForwardTransformerOracle(cfg, W, torch.float64)
    val = ref.val_step(*batch)
    ref.learning_rate = 1e-3
    tr = ref.train_step(*batch)
    out = {f'w::{k}': v.astype(np.float32) for k, v in W.items()}
    out.update(tokens=batch[0], mel_target=batch[1], durations=batch[2], pitch=batch[3])
    out.update(val_mel=val['mel'].numpy().astype(np.float32), val_duration=val['duration'].numpy(),
               val_pitch=val['pitch'].numpy(), val_loss=np.float64(val['loss']),
               val_losses=np.array([float(val['losses'][k]) for k in ('mel', 'duration', 'pitch')]),
               val_expanded_mask=val['expanded_mask'].numpy().astype(np.float32),
               val_dec_attn_last=val['decoder_attention']['Decoder_DenseBlock2_SelfAttention'][1].numpy().astype(np.float32),
               train_loss=np.float64(tr['loss']))
    for k, g in tr['grads'].items():
        out[f'gabs::{k}'] = np.float64(g.abs().max())
        out[f'gsum::{k}'] = np.float64(g.sum())
    for k, v in ref.weights_numpy().items():          # post-Adam weights: 4 tensors in full, sums of all
        out[f'w1sum::{k}'] = np.float64(v.sum())
        if k in ('out.w', 'dec.blk1.ffn.w2', 'enc.blk0.wq', 'dur.conv1.w'):
            out[f'w1::{k}'] = v.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, 'ft_tiny.npz'), **out)

    # ---- length regulator ------------------------------------------------------------------------
    rng = np.random.default_rng(5)
    dur = rng.choice([0., 0.5, 1.5, 2.5, 3.5, 1., 2., 4.49, 4.5, 6., 0.49999997], size=(5, 40)).astype(np.float32)
    dur[3] = 0
    dur[4, 10:] = 0
    idx, lens, out_len = fo.expand_indices_np(dur[..., None])
    np.savez_compressed(os.path.join(HERE, 'lenreg.npz'), dur=dur, idx=idx, lens=lens, out_len=np.int32(out_len))

    # ---- mel -------------------------------------------------------------------------------------
    clips = [mo.synthetic_clip(24000, seed=1), mo.synthetic_clip(25600, seed=2)]
    np.savez_compressed(os.path.join(HERE, 'mel.npz'), clip0=clips[0], clip1=clips[1],
                        mel0=mo.mel_spectrogram(clips[0]), mel1=mo.mel_spectrogram(clips[1]))
    for f in ('ft_tiny.npz', 'lenreg.npz', 'mel.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
