"""Phoneme symbol table - data, not code: the vocabulary the released checkpoints were trained with
(reference data/text/symbols.py:1-12).  126 symbols -> vocab 127 with the pad id 0."""
_vowels = 'iyɨʉɯuɪʏʊeøɘəɵɤoɛœɜɞʌɔæɐaɶɑɒᵻ'
_non_pulmonic_consonants = 'ʘɓǀɗǃʄǂɠǁʛ'
_pulmonic_consonants = 'pbtdʈɖcɟkɡqɢʔɴŋɲɳnɱmʙrʀⱱɾɽɸβfvθðszʃʒʂʐçʝxɣχʁħʕhɦɬɮʋɹɻjɰlɭʎʟ'
_suprasegmentals = 'ˈˌːˑ'
_other_symbols = 'ʍwɥʜʢʡɕʑɺɧ'
_diacrilics = 'ɚ˞ɫ'
_punctuations = '!,-.:;? \'()'

_phonemes = sorted(set(_vowels + _non_pulmonic_consonants + _pulmonic_consonants + _suprasegmentals
                       + _other_symbols + _diacrilics))
all_phonemes = sorted(list(_phonemes) + list(_punctuations))
