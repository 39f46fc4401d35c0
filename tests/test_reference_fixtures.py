"""The oracle AND the product's host-side functions against vectors produced by the reference's OWN code
(tests/golden/make_reference_fixtures.py: the NumPy / pure-Python parts of the reference run here with empty
stand-ins for the third-party imports they do not use).  These rows are pinned; the TensorFlow / librosa
arithmetic is not."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ft_oracle as fo
from oracle import mel_oracle as mo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def ref():
    with np.load(os.path.join(HERE, 'golden', 'reference_numpy_fixtures.npz')) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='module')
def tok():
    with open(os.path.join(HERE, 'golden', 'reference_tokenizer_fixtures.json'), encoding='utf8') as f:
        return json.load(f)


def test_positional_encoding_equals_the_reference_bit_for_bit(ref):
    """SURVEY 8 row a5 (model/transformer_utils.py:5-21)."""
    from transformertts_amd.model.transformer_utils import positional_encoding
    for impl in (fo.positional_encoding, positional_encoding):
        np.testing.assert_array_equal(impl(64, 32), ref['pe_64x32'])
        np.testing.assert_array_equal(impl(2000, 256)[ref['pe_enc_rows']], ref['pe_enc_2000x256_rows'])
        np.testing.assert_array_equal(impl(10000, 256)[ref['pe_dec_rows']], ref['pe_dec_10000x256_rows'])
        np.testing.assert_array_equal(impl(10000, 384)[ref['pe_dec_rows']], ref['pe_dec_10000x384_rows'])
    assert ref['pe_64x32'].dtype == np.float32


def test_padding_masks_equal_the_reference(ref):
    """Row a4 (transformer_utils.py:24-32), oracle side; the device kernels are compared with the oracle in
    tests/test_ops_gpu.py::test_masks_embedding_pitch_rowdot."""
    m = fo.create_encoder_padding_mask(torch.from_numpy(ref['mask_tokens']), torch.float32).numpy()
    np.testing.assert_array_equal(m, ref['mask_enc'])
    assert m.shape == (3, 1, 1, 9)
    m = fo.create_mel_padding_mask(torch.from_numpy(ref['mask_mel_in'])).numpy()
    np.testing.assert_array_equal(m, ref['mask_mel'])
    assert ref['mask_mel'][1, 0, 0, 2] == 1.0           # content-derived: an all-zero frame counts as padding


def test_normalisers_equal_the_reference(ref):
    """Row a19, the normaliser step (data/audio.py:209-242): oracle normalize, product denormalize."""
    from transformertts_amd.data.audio import MelGAN, WaveRNN
    S = ref['norm_in']
    np.testing.assert_array_equal(mo.melgan_normalize(S), ref['melgan_norm'])
    np.testing.assert_array_equal(mo.wavernn_normalize(S), ref['wavernn_norm'])
    np.testing.assert_array_equal(MelGAN().denormalize(ref['melgan_norm']), ref['melgan_denorm'])
    np.testing.assert_array_equal(WaveRNN().denormalize(ref['wavernn_norm']), ref['wavernn_denorm'])
    np.testing.assert_array_equal(WaveRNN().denormalize(np.array([-7.0, -4.0, 0.0, 4.0, 9.0])),
                                  ref['wavernn_denorm_out_of_range'])
    assert ref['melgan_norm'][6, 0] == np.log(np.float32(1e-5))


def test_wav_preprocessing_equals_the_reference(ref):
    """SURVEY 8f.4 (data/audio.py:132-141,154-162)."""
    from transformertts_amd.data.audio import normalize_volume, pad_for_frame_count
    q, l = ref['wav_quiet'], ref['wav_loud']
    np.testing.assert_array_equal(normalize_volume(q, -30, 32767, increase_only=True), ref['wav_quiet_normvol_inc'])
    np.testing.assert_array_equal(normalize_volume(l, -30, 32767, increase_only=True), ref['wav_loud_normvol_inc'])
    np.testing.assert_array_equal(normalize_volume(l, -30, 32767, decrease_only=True), ref['wav_loud_normvol_dec'])
    np.testing.assert_array_equal(normalize_volume(l, -30, 32767), ref['wav_loud_normvol'])
    pre = pad_for_frame_count(normalize_volume(q, -30, 32767, increase_only=True), 256)
    np.testing.assert_array_equal(pre, ref['wav_quiet_preprocessed'])
    assert pre.shape == (256 * 6 + 1,) and pre.dtype == ref['wav_quiet_preprocessed'].dtype
    np.testing.assert_array_equal(pad_for_frame_count(normalize_volume(l, -30, 32767, increase_only=True), 256),
                                  ref['wav_loud_preprocessed'])
    np.testing.assert_array_equal(pad_for_frame_count(q, 256), ref['wav_quiet_preprocessed_nonorm'])


def test_schedules_equal_the_reference(ref):
    """utils/scheduling.py:5-48 (the caller side of `set_constants`, train_tts.py:152-153)."""
    from transformertts_amd.utils.scheduling import piecewise_linear_schedule, reduction_schedule
    table = ref['sched_lr_table'].tolist()
    got = np.array([piecewise_linear_schedule(int(s), table) for s in ref['sched_steps']])
    np.testing.assert_array_equal(got, ref['sched_lr_f32'])
    assert got.dtype == np.float32
    red = ref['sched_red_table'].tolist()
    np.testing.assert_array_equal([reduction_schedule(int(s), red) for s in ref['sched_red_steps']], ref['sched_red'])


def test_tokenizer_equals_the_reference(tok):
    """data/text/tokenizer.py:9-46 + symbols.py: the vocabulary the model's Embedding is sized by."""
    from transformertts_amd.data.text import Tokenizer
    t = Tokenizer(add_start_end=False, model_breathing=False)
    assert t.alphabet == tok['all_phonemes'] and t.vocab_size == tok['vocab_size_model'] == fo.VOCAB_SIZE
    assert t(tok['sentence']) == tok['encode_model'] and t.decode(t(tok['sentence'])) == tok['decode_model']
    t2 = Tokenizer()
    assert t2.vocab_size == tok['vocab_size_default']
    assert t2(tok['sentence']) == tok['encode_default'] and t2.decode(t2(tok['sentence'])) == tok['decode_default']
    assert [t2.start_token_index, t2.end_token_index, t2.breathing_token_index] == tok['start_end_breathing_default']
    t3 = Tokenizer(alphabet=list('ab c'))
    assert t3.alphabet == tok['abc_alphabet'] and t3.vocab_size == tok['abc_vocab_size']
    assert t3('a b c') == tok['abc_encode'] and t3.decode(t3('a b c')) == tok['abc_decode']
