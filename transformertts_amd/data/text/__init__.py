"""Token-id tables for the phoneme vocabulary (the only part of the reference's text front end that is on
the hot path: ForwardTransformer's Embedding has one row per id, reference model/models.py:375-383).

The id space is a flat table, built once from `symbols.VOCABULARY`:

    id 0                      padding
    ids 1 .. V                the V symbols of the alphabet, in code-point order
    ids V+1, V+2              start / end markers          (only with add_start_end)
    next id                   breathing marker '@'          (only with model_breathing)

`Tokenizer(text)` is then a table lookup per character; with `model_breathing` a space maps to the PAIR
(space id, breathing id) and every sentence opens with the breathing id.  Attribute names are the ones the
reference's callers read (data/text/tokenizer.py:9-47: alphabet, vocab_size, idx_to_token, token_to_idx,
start/end/breathing_token_index); tests/test_reference_fixtures.py::test_tokenizer_equals_the_reference
holds ids, decode() and vocab sizes to the reference's own tokenizer for every flag combination.

Grapheme -> phoneme conversion (espeak through the third-party `phonemizer` package) is OUT OF SCOPE
(SURVEY.md section 2): callers pass phoneme strings or token ids (`predict(..., encode=False)`)."""
from __future__ import annotations

from typing import Callable, List, Optional

from .symbols import VOCABULARY


class Tokenizer:
    def __init__(self, start_token='>', end_token='<', pad_token='/', add_start_end=True, alphabet=None,
                 model_breathing=True):
        self.alphabet = sorted(set(alphabet)) if alphabet else list(VOCABULARY)
        self.add_start_end, self.model_breathing = bool(add_start_end), bool(model_breathing)
        self.breathing_token = '@'
        table = [pad_token, *self.alphabet]                     # table[id] = printable symbol of that id
        if self.add_start_end:
            self.start_token_index, self.end_token_index = len(table), len(table) + 1
            table += [start_token, end_token]
        if self.model_breathing:
            self.breathing_token_index = len(table)
            table.append(self.breathing_token)
        self.vocab_size = len(table)
        self.idx_to_token = dict(enumerate(table))
        # symbol -> the ids one occurrence expands to (start / end markers are positions, not input symbols)
        typed = range(len(self.alphabet) + 1)
        self.token_to_idx = {table[i]: [i] for i in typed}
        if self.model_breathing:
            self.token_to_idx[self.breathing_token] = [self.breathing_token_index]
            self.token_to_idx[' '] = self.token_to_idx[' '] + [self.breathing_token_index]

    def __call__(self, sentence: str) -> List[int]:
        ids: List[int] = [self.breathing_token_index] if self.model_breathing else []
        for ch in sentence:
            ids.extend(self.token_to_idx[ch])                   # an unknown symbol is a KeyError, as in the reference
        return [self.start_token_index, *ids, self.end_token_index] if self.add_start_end else ids

    def decode(self, sequence) -> str:
        return ''.join(self.idx_to_token[int(t)] for t in sequence)


class TextToTokens:
    """`model.text_pipeline` (reference data/text/__init__.py:7-21): phonemes -> ids.  `phonemizer` is any
    callable text -> phoneme string the application supplies; none is bundled."""

    def __init__(self, phonemizer: Optional[Callable[[str], str]], tokenizer: Tokenizer):
        self.phonemizer, self.tokenizer = phonemizer, tokenizer

    def __call__(self, input_text):
        if self.phonemizer is None:
            raise RuntimeError('no phonemizer is attached (espeak / `phonemizer` are outside this package): set '
                               'model.text_pipeline.phonemizer to a callable, pass a phoneme string to '
                               'model.text_pipeline.tokenizer, or call predict(ids, encode=False)')
        return self.tokenizer(self.phonemizer(input_text))

    @classmethod
    def default(cls, language: str, add_start_end: bool, with_stress: bool, model_breathing: bool, njobs=1,
                phonemizer=None):
        return cls(phonemizer, Tokenizer(add_start_end=add_start_end, model_breathing=model_breathing))
