#!/usr/bin/env python
"""Runs the reference's OWN ForwardTransformer source (model/models.py, model/layers.py,
model/transformer_utils.py, utils/losses.py imported from /root/reference) over the torch-float64 stand-in for
TensorFlow in tests/_tf_shim.py, on seeded weights and a ragged batch, and freezes what it computes:
forward outputs, the three losses and their weighted sum, and the gradient of every variable.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_source_run.py

-> tests/golden/reference_source_run.npz (committed; the GPU box and later checkouts have no /root/reference).
See tests/_tf_shim.py for what this pins (the reference's wiring, executed) and what it does not
(TensorFlow's floating point)."""
import os
import sys

sys.dont_write_bytecode = True          # never write __pycache__ into the read-only reference tree

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('TTS_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

CASES = {
    # name: (oracle make_config kwargs, batch B, Tp, Tm, seeds)
    # (head dims of 32 so that the GPU kernels can run the same cases: tests/test_zz_reference_source_gpu.py)
    'dense': dict(cfg=dict(d_model=64, enc_heads=(2,), dec_heads=(2,), ffn=64, dur_filters=(24, 16),
                           pitch_filters=(24, 16)), B=3, Tp=14, Tm=47, wseed=101, bseed=7),
    'conv': dict(cfg=dict(d_model=32, enc_heads=(1, 1), dec_heads=(1, 1), ffn=48, enc_dense_blocks=1,
                          dec_dense_blocks=1, conv_filters=(40, 32), conv_kernel=3, dur_filters=(24, 16),
                          pitch_filters=(24, 16)), B=2, Tp=11, Tm=38, wseed=202, bseed=8),
}


def build_reference_model(cfg, W):
    """Construct the reference class, build its variables with one call, then assign `W` (this package's
    variable names) through the Keras order table - which is thereby checked against the reference
    constructors' real attribute order (every slot's shape must match)."""
    from model.models import ForwardTransformer
    from oracle import ft_oracle as fo
    from transformertts_amd.model.keras_weights import keras_layer_table
    m = ForwardTransformer(**cfg)
    import _tf_shim
    m._compile(optimizer=_tf_shim.RecordingOptimizer())
    m.call(torch.ones((1, 2), dtype=torch.int64))             # builds every layer (reference: build_model_weights)
    table = keras_layer_table(cfg, fo.VOCAB_SIZE)
    layers = m.layers
    assert [n for n, _ in table][:5] == [l.name for l in layers][:5], [l.name for l in layers]
    assert len(table) == len(layers)
    with torch.no_grad():
        for (lname, entries), layer in zip(table, layers):
            ws = layer.weights
            assert len(ws) == len(entries), (lname, len(ws), len(entries))
            for (kname, ref_name, shape), w in zip(entries, ws):
                assert tuple(w.shape) == tuple(shape), (lname, kname, tuple(w.shape), shape)
                w.copy_(torch.from_numpy(np.asarray(W[ref_name], dtype=np.float64)).reshape(w.shape))
    return m


def run_case(name):
    from oracle import ft_oracle as fo
    c = CASES[name]
    cfg = fo.make_config(**c['cfg'])
    W = fo.init_weights(cfg, seed=c['wseed'], perturb=0.05)
    tok, mel, dur, pit = fo.synthetic_batch(c['B'], c['Tp'], c['Tm'], seed=c['bseed'], ragged=True)
    m = build_reference_model(cfg, W)
    t = lambda a: torch.from_numpy(np.asarray(a))
    out = m._train_step(t(tok), t(mel).double(), t(dur), t(pit).double())          # dropout 0: deterministic
    from transformertts_amd.model.keras_weights import keras_layer_table
    names = [r for _, es in keras_layer_table(cfg, fo.VOCAB_SIZE) for _, r, _ in es]
    vars_ = [w for l in m.layers for w in l.weights]
    applied = m.optimizer.applied                       # (gradient, variable) pairs of the reference's own step
    assert [id(v) for _, v in applied] == [id(v) for v in vars_]
    grads = [g for g, _ in applied]
    res = {'tokens': tok, 'mel_target': mel, 'durations': dur, 'pitch': pit,
           'out_mel': out['mel'].detach().numpy(), 'out_duration': out['duration'].detach().numpy(),
           'out_pitch': out['pitch'].detach().numpy(), 'out_expanded_mask': out['expanded_mask'].detach().numpy(),
           'loss': np.float64(out['loss']), 'losses': np.array([float(out['losses'][k]) for k in ('mel', 'duration', 'pitch')])}
    enc_key = sorted(out['encoder_attention'])[-1]
    dec_key = sorted(out['decoder_attention'])[-1]
    res['enc_attn_key'], res['dec_attn_key'] = enc_key, dec_key
    res['enc_attn'] = out['encoder_attention'][enc_key].detach().numpy()
    res['dec_attn'] = out['decoder_attention'][dec_key].detach().numpy()
    for n, g in zip(names, grads):
        res[f'grad::{n}'] = np.zeros(()) if g is None else g.detach().numpy()
    # inference through the reference's predict() (models.py:559-595): predicted durations drive the length
    # regulator, scaled by 1/speed_regulator and clamped per symbol.  Untrained weights predict ~0 frames, so the
    # duration head's bias is shifted; the clamped symbols are the first two distinct tokens of the sentence.
    row = np.asarray(tok[0][tok[0] > 0])
    other = int(next(x for x in row if x != row[0]))
    for shift in (2.3, 2.6, 2.9, 3.2, 1.9, 3.5):               # first shift that keeps every duration off a .5 tie
        W2 = dict(W)
        W2['dur.lin.b'] = W['dur.lin.b'] + shift
        m2 = build_reference_model(cfg, W2)
        idx_to_token = m2.text_pipeline.tokenizer.idx_to_token
        sym_max, sym_min = idx_to_token[int(row[0])], idx_to_token[other]
        pred = m2.predict(t(row), encode=False, speed_regulator=0.8, phoneme_max_duration={sym_max: 2.0},
                          phoneme_min_duration={sym_min: 4.0})
        use = np.asarray(pred['duration'].detach().numpy()).reshape(-1) / 0.8
        use = np.where(row == row[0], np.minimum(use, 2.0), use)
        use = np.where(row == other, np.maximum(use, 4.0), use)
        if np.abs(np.abs(use - np.floor(use)) - 0.5).min() > 0.03:
            break
    else:
        raise SystemExit('every candidate shift leaves a duration on a rounding tie')
    res.update(pred_tokens=row, pred_sym_max=sym_max, pred_sym_min=sym_min, pred_bias_shift=np.float64(shift),
               pred_mel=pred['mel'].detach().numpy(), pred_duration=pred['duration'].detach().numpy())
    return res


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f'{REF} not found: this fixture can only be regenerated next to the reference')
    import _tf_shim
    _tf_shim.install()
    sys.path.insert(0, REF)
    out = {}
    for name in CASES:
        for k, v in run_case(name).items():
            out[f'{name}/{k}'] = v
    np.savez_compressed(os.path.join(HERE, 'reference_source_run.npz'), **out)
    print('wrote reference_source_run.npz:', len(out), 'arrays;',
          {n: float(out[f'{n}/loss']) for n in CASES})


if __name__ == '__main__':
    main()
