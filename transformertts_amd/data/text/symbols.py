"""The phoneme vocabulary in token-id order - data, not code.

The released checkpoints' Embedding has one row per symbol of the reference's inventory (reference
data/text/symbols.py:1-12: IPA vowels, pulmonic and non-pulmonic consonants, suprasegmentals, a few extra
symbols and diacritics, plus punctuation, sorted by code point) and row 0 for padding.  It is stored here the
way this package uses it: ONE string whose character at position i is the symbol of token id i + 1, so the
vocabulary size (len + 1 = 127) and every id can be read off directly.
tests/test_reference_fixtures.py::test_tokenizer_equals_the_reference holds it to the reference's own list."""

VOCABULARY = (
    " !'(),-.:;?abcdefhijklmnopqrstuvwxyzæ"
    'çðøħŋœǀǁǂǃɐɑɒɓɔɕɖɗɘəɚɛɜɞɟɠɡɢɣɤɥɦɧɨɪɫɬɭɮɯɰɱɲɳɴɵɶɸ'
    'ɹɺɻɽɾʀʁʂʃʄʈʉʊʋʌʍʎʏʐʑʒʔʕʘʙʛʜʝʟʡʢˈˌːˑ˞βθχᵻⱱ'
)
PUNCTUATION = " !'(),-.:;?"          # the punctuation subset of the vocabulary

all_phonemes = list(VOCABULARY)       # the reference's name for the same list (data/text/symbols.py:12)
assert len(set(VOCABULARY)) == len(VOCABULARY) == 126 and all_phonemes == sorted(all_phonemes)
