#!/usr/bin/env python
"""Golden vectors produced by the REFERENCE'S OWN CODE, for the parts of the path that are plain NumPy /
Python in the reference and therefore run here although TensorFlow, librosa, phonemizer ... are not
installable: the modules are imported from /root/reference with empty stand-ins for the third-party
imports they do not touch in these functions.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_fixtures.py     # needs /root/reference

What runs unmodified reference code (NumPy only):
  * `model/transformer_utils.py:5-21`  get_angles / positional_encoding  (the trailing `tf.cast(x, tf.float32)`
    is the stand-in's `np.asarray(x, np.float32)`)
  * `data/audio.py:154-162,132-141`    Audio.normalize_volume / Audio.preprocess (volume normalisation + the
    one-sample pad; the trimming flags off)
  * `data/audio.py:209-242`            MelGAN / WaveRNN normalize + denormalize
  * `utils/scheduling.py:5-48`         piecewise_linear_schedule / reduction_schedule
  * `data/text/tokenizer.py:9-46`, `data/text/symbols.py`   Tokenizer (alphabet, breathing / start / end tokens)
What runs reference code over a five-op NumPy stand-in for TensorFlow (`tf.cast`, `tf.math.equal`,
`tf.math.abs`, `tf.reduce_sum`, `tf.newaxis` - each the obvious NumPy call):
  * `model/transformer_utils.py:24-32`  create_encoder_padding_mask / create_mel_padding_mask
Everything that needs real TensorFlow / librosa arithmetic (layers, losses, Adam, STFT, mel basis) stays
"parity unpinned" (DESIGN.md section 2).

Output: tests/golden/reference_numpy_fixtures.npz + reference_tokenizer_fixtures.json (committed; the GPU
box has no /root/reference)."""
import json
import os
import sys
import types

sys.dont_write_bytecode = True          # never write __pycache__ into the read-only reference tree

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('TTS_REFERENCE', '/root/reference')


def _stand_ins():
    tf = types.ModuleType('tensorflow')
    tf.float32, tf.int32, tf.newaxis = np.float32, np.int32, np.newaxis
    tf.cast = lambda x, dtype=None, **kw: np.asarray(x, dtype=kw.get('dtype', dtype))
    tf.reduce_sum = lambda x, axis=None: np.sum(x, axis=axis)
    tf.abs = np.abs
    tf.math = types.SimpleNamespace(equal=lambda a, b: np.equal(a, b), abs=np.abs)
    sys.modules['tensorflow'] = tf
    for name in ('librosa', 'librosa.display', 'matplotlib', 'matplotlib.pyplot', 'soundfile', 'webrtcvad',
                 'pyworld', 'phonemizer', 'phonemizer.phonemize'):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules['matplotlib'].pyplot = sys.modules['matplotlib.pyplot']
    sys.modules['librosa'].display = sys.modules['librosa.display']
    sys.modules['phonemizer'].phonemize = sys.modules['phonemizer.phonemize']
    sys.modules['phonemizer.phonemize'].phonemize = None          # imported by name, never called here


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f'{REF} not found: the fixtures can only be regenerated next to the reference')
    _stand_ins()
    sys.path.insert(0, REF)
    from model import transformer_utils as tu
    from data import audio as ra
    from utils import scheduling as sch
    from data.text.tokenizer import Tokenizer
    from data.text import symbols

    out = {}
    # ---- positional encoding: a small table in full, and sampled rows of the two production tables
    out['pe_64x32'] = np.asarray(tu.positional_encoding(64, 32))[0]
    rows = np.array([0, 1, 2, 3, 10, 777, 1999])
    out['pe_enc_rows'] = rows
    out['pe_enc_2000x256_rows'] = np.asarray(tu.positional_encoding(2000, 256))[0][rows]
    rows = np.array([0, 1, 899, 900, 4096, 9999])
    out['pe_dec_rows'] = rows
    out['pe_dec_10000x256_rows'] = np.asarray(tu.positional_encoding(10000, 256))[0][rows]
    out['pe_dec_10000x384_rows'] = np.asarray(tu.positional_encoding(10000, 384))[0][rows]

    # ---- padding masks (reference code over the five-op stand-in)
    rng = np.random.default_rng(11)
    tok = rng.integers(1, 127, size=(3, 9)).astype(np.int32)
    tok[0, 6:] = 0
    tok[2, :] = 0
    out['mask_tokens'] = tok
    out['mask_enc'] = np.asarray(tu.create_encoder_padding_mask(tok))
    x = rng.standard_normal((2, 7, 5)).astype(np.float32)
    x[0, 4:] = 0
    x[1, 2, :] = 0                                   # an all-zero row in the middle counts as padding too
    out['mask_mel_in'] = x
    out['mask_mel'] = np.asarray(tu.create_mel_padding_mask(x))

    # ---- normalisers
    S = np.abs(rng.standard_normal((80, 23))).astype(np.float32) * 3
    S[5, :4] = 1e-7                                  # below both clip floors
    S[6, 0] = 0.0
    out['norm_in'] = S
    g, w = ra.MelGAN(), ra.WaveRNN()
    out['melgan_norm'], out['wavernn_norm'] = g.normalize(S), w.normalize(S)
    out['melgan_denorm'] = g.denormalize(out['melgan_norm'])
    out['wavernn_denorm'] = w.denormalize(out['wavernn_norm'])
    out['wavernn_denorm_out_of_range'] = w.denormalize(np.array([-7.0, -4.0, 0.0, 4.0, 9.0]))

    # ---- wav preprocessing
    cfg = dict(sampling_rate=22050, n_fft=1024, mel_channels=80, hop_length=256, win_length=1024, f_min=0,
               f_max=8000, normalizer='MelGAN', norm_wav=True, target_dBFS=-30, int16_max=32767,
               trim_long_silences=False, trim_silence=False)
    au = ra.Audio(**cfg)
    quiet = (0.003 * rng.standard_normal(256 * 6)).astype(np.float32)
    loud = (0.4 * rng.standard_normal(1000)).astype(np.float32)
    out['wav_quiet'], out['wav_loud'] = quiet, loud
    out['wav_quiet_normvol_inc'] = au.normalize_volume(quiet, increase_only=True)
    out['wav_loud_normvol_inc'] = au.normalize_volume(loud, increase_only=True)
    out['wav_loud_normvol_dec'] = au.normalize_volume(loud, decrease_only=True)
    out['wav_loud_normvol'] = au.normalize_volume(loud)
    out['wav_quiet_preprocessed'] = au.preprocess(quiet)            # 1536 = 6 hops -> padded to 1537
    out['wav_loud_preprocessed'] = au.preprocess(loud)              # 1000 % 256 != 0 -> untouched length
    au2 = ra.Audio(**dict(cfg, norm_wav=False))
    out['wav_quiet_preprocessed_nonorm'] = au2.preprocess(quiet)

    # ---- schedules
    lr_sched = [[0, 1.0e-4], [40000, 5.0e-5], [100000, 1.0e-5]]
    steps = np.array([0, 1, 39999, 40000, 70000, 100000, 250000])
    out['sched_steps'] = steps
    out['sched_lr_table'] = np.array(lr_sched)
    out['sched_lr'] = np.array([float(sch.piecewise_linear_schedule(int(s), lr_sched)) for s in steps], np.float64)
    out['sched_lr_f32'] = np.array([np.asarray(sch.piecewise_linear_schedule(int(s), lr_sched)) for s in steps])
    red = [[0, 10], [80000, 5], [100000, 2], [130000, 1]]
    out['sched_red_table'] = np.array(red)
    rsteps = np.array([0, 79999, 80000, 99999, 100000, 129999, 130000, 10 ** 6])
    out['sched_red_steps'] = rsteps
    out['sched_red'] = np.array([sch.reduction_schedule(int(s), red) for s in rsteps])
    np.savez_compressed(os.path.join(HERE, 'reference_numpy_fixtures.npz'), **out)

    # ---- tokenizer
    t = Tokenizer(add_start_end=False, model_breathing=False)       # what ForwardTransformer builds (models.py:373-377)
    t2 = Tokenizer()                                                # defaults: start/end + breathing
    t3 = Tokenizer(alphabet=list('ab c'))                           # tests/test_char_tokenizer.py
    sent = 'həloʊ wɜːld, ðɪs ɪz ɐ tɛst.'
    tok_out = {
        'all_phonemes': list(symbols.all_phonemes),
        'vocab_size_model': t.vocab_size, 'vocab_size_default': t2.vocab_size, 'sentence': sent,
        'encode_model': t(sent), 'encode_default': t2(sent), 'decode_model': t.decode(t(sent)),
        'decode_default': t2.decode(t2(sent)),
        'start_end_breathing_default': [t2.start_token_index, t2.end_token_index, t2.breathing_token_index],
        'abc_alphabet': t3.alphabet, 'abc_vocab_size': t3.vocab_size, 'abc_encode': t3('a b c'),
        'abc_decode': t3.decode(t3('a b c')),
    }
    with open(os.path.join(HERE, 'reference_tokenizer_fixtures.json'), 'w', encoding='utf8') as f:
        json.dump(tok_out, f, ensure_ascii=False, indent=1)
    print('wrote reference_numpy_fixtures.npz', {k: np.shape(v) for k, v in list(out.items())[:6]}, '...')
    print('wrote reference_tokenizer_fixtures.json: vocab', t.vocab_size, t2.vocab_size)


if __name__ == '__main__':
    main()
