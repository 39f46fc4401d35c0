"""Synthetic LJSpeech-shaped inputs and the benchmark configurations.
There is no network for datasets or checkpoints: benchmarks use these seeded tensors and
randomly initialised weights of the named architecture."""
from __future__ import annotations

import numpy as np


def make_config(d_model=256, enc_heads=(4,) * 6, dec_heads=(4,) * 6, ffn=1024, enc_dense_blocks=None,
                dec_dense_blocks=None, conv_filters=None, conv_kernel=3, dur_filters=(256, 226),
                pitch_filters=(256, 226), dur_kernel=3, pitch_kernel=3, mel_channels=80, enc_max_pos=2000,
                dec_max_pos=10000, dropout_rate=0.1, predictors_dropout=0.1) -> dict:
    """Flat config with the reference constructor keywords (model/models.py:345-372).  Defaults =
    BASELINE.json configs[1]: d_model 256, 6+6 self-attention DENSE blocks, 4 heads, FFN 1024
    (the FFN width is not given by BASELINE.json - SURVEY.md section 0.4)."""
    enc_heads, dec_heads = list(enc_heads), list(dec_heads)
    return dict(
        encoder_model_dimension=d_model, decoder_model_dimension=d_model, dropout_rate=dropout_rate,
        decoder_num_heads=dec_heads, encoder_num_heads=enc_heads,
        encoder_max_position_encoding=enc_max_pos, decoder_max_position_encoding=dec_max_pos,
        encoder_dense_blocks=len(enc_heads) if enc_dense_blocks is None else enc_dense_blocks,
        decoder_dense_blocks=len(dec_heads) if dec_dense_blocks is None else dec_dense_blocks,
        duration_conv_filters=list(dur_filters), pitch_conv_filters=list(pitch_filters),
        duration_kernel_size=dur_kernel, pitch_kernel_size=pitch_kernel,
        predictors_dropout=predictors_dropout, mel_channels=mel_channels, phoneme_language='en-us',
        with_stress=True, model_breathing=False, transposed_attn_convs=True,
        encoder_attention_conv_filters=None if conv_filters is None else list(conv_filters),
        decoder_attention_conv_filters=None if conv_filters is None else list(conv_filters),
        encoder_attention_conv_kernel=conv_kernel, decoder_attention_conv_kernel=conv_kernel,
        encoder_feed_forward_dimension=ffn, decoder_feed_forward_dimension=ffn)


def synthetic_batch(B: int, Tp: int, Tm: int, mel_channels: int = 80, seed: int = 1234,
                    ragged: bool = False, vocab_size: int = 127):
    """(tokens i32 [B,Tp], mel f32 [B,Tm,C], durations i32 [B,Tp], pitch f32 [B,Tp]).
    max-shape set: every sample Tp phonemes / Tm frames, durations = multinomial(Tm, uniform) (zeros
    allowed, sum == Tm), pitch ~ N(0,1) with 30 % exact zeros, mel ~ clip(N(-5,2), -11.5129, 2).
    ragged=True: per-sample lengths, zero padded, sample 0 at both maxima, sum(dur_b) == mel_len_b."""
    rng = np.random.default_rng(seed)
    tokens = np.zeros((B, Tp), dtype=np.int32)
    durs = np.zeros((B, Tp), dtype=np.int32)
    pitch = np.zeros((B, Tp), dtype=np.float32)
    mel = np.zeros((B, Tm, mel_channels), dtype=np.float32)
    for b in range(B):
        if ragged and b > 0:
            tp = int(rng.integers(max(1, (3 * Tp) // 10), Tp + 1))
            tm = int(min(Tm, max(tp, rng.integers(max(1, (3 * Tm) // 10), Tm + 1))))
        else:
            tp, tm = Tp, Tm
        tokens[b, :tp] = rng.integers(1, vocab_size, size=tp)
        durs[b, :tp] = rng.multinomial(tm, np.full(tp, 1.0 / tp))
        p = rng.standard_normal(tp).astype(np.float32)
        p[rng.random(tp) < 0.3] = 0.0
        pitch[b, :tp] = p
        mel[b, :tm] = np.clip(rng.normal(-5.0, 2.0, size=(tm, mel_channels)), -11.5129, 2.0)
    return tokens, mel, durs, pitch


def learnable_batch(B: int, Tp: int, Tm: int, mel_channels: int = 80, seed: int = 1234, vocab_size: int = 127):
    """A ragged batch whose TARGETS ARE FUNCTIONS OF ITS INPUTS, so that a few hundred optimiser steps can fit it (the
    noise targets of `synthetic_batch` cannot be: their loss plateaus at the noise's mean absolute deviation): the duration
    of a phoneme is 1 + token % 7, its pitch a fixed per-token value, and every mel frame the fixed 80-bin row of the token
    it expands from.  Sample 0 has Tp phonemes and exactly Tm frames (durations rescaled), the others are shorter;
    sum(dur_b) == mel_len_b.  Returns (tokens i32 [B,Tp], mel f32 [B,Tm,C], durations i32 [B,Tp], pitch f32 [B,Tp])."""
    rng = np.random.default_rng(seed)
    table = np.clip(rng.normal(-5.0, 2.0, size=(vocab_size, mel_channels)), -11.5129, 2.0).astype(np.float32)
    ptab = rng.standard_normal(vocab_size).astype(np.float32)
    tok = np.zeros((B, Tp), np.int32)
    dur = np.zeros((B, Tp), np.int32)
    pit = np.zeros((B, Tp), np.float32)
    mel = np.zeros((B, Tm, mel_channels), np.float32)
    for b in range(B):
        tp = Tp if b == 0 else int(rng.integers(max(1, Tp // 2), Tp + 1))
        t = rng.integers(1, vocab_size, size=tp)
        d = 1 + (t % 7)
        if b == 0:                                     # stretch to exactly Tm frames
            d = np.maximum(1, np.floor(d * (Tm / d.sum()))).astype(np.int64)
            d[np.argmax(d)] += Tm - d.sum()
            assert d.min() >= 1 and d.sum() == Tm
        while d.sum() > Tm:                            # shorter samples: drop phonemes until the frames fit
            tp -= 1
            t, d = t[:tp], d[:tp]
        tok[b, :tp], dur[b, :tp], pit[b, :tp] = t, d, ptab[t]
        mel[b, :int(d.sum())] = np.repeat(table[t], d, axis=0)
    return tok, mel, dur, pit
