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

Grapheme -> phoneme conversion (espeak through the third-party `phonemizer` package) is OUT OF SCOPE as a
component and neither dependency exists offline; `Phonemizer` is a thin adapter that keeps the
reference's text entry point working WHERE THEY ARE INSTALLED (`model.predict("some text")`, reference
data/text/tokenizer.py:50-104) and raises a clear error where they are not.  Callers without espeak pass phoneme
strings to the tokenizer or token ids to `predict(..., encode=False)`."""
from __future__ import annotations

import re
from typing import Callable, List, Optional, Union

from .symbols import PUNCTUATION, VOCABULARY


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


class Phonemizer:
    """text -> IPA phoneme string through `phonemizer.phonemize` (espeak backend), imported on first use.  Same
    constructor and call signature as the reference's class (data/text/tokenizer.py:50-76).  Around the third-party
    call: '-' travels as an em dash (espeak drops hyphens), symbols outside the model's vocabulary are removed from
    the result, runs of white space collapse and white space next to punctuation disappears."""
    _EM_DASH = '\u2014'
    _MARKS = ';:,.!?\u00a1\u00bf\u2014\u2026"\u00ab\u00bb\u201c\u201d'

    def __init__(self, language: str, with_stress: bool, njobs=4):
        self.language, self.with_stress, self.njobs = language, with_stress, njobs
        self._known = frozenset(VOCABULARY)
        self._spaces = re.compile(r'\s+')
        self._around_marks = re.compile(r'\s*([' + re.escape(PUNCTUATION.strip()) + r'])\s*')

    def _clean(self, phonemes: str) -> str:
        kept = ''.join(ch for ch in phonemes.replace(self._EM_DASH, '-') if ch in self._known)
        return self._around_marks.sub(r'\1', self._spaces.sub(' ', kept)).strip()

    def __call__(self, text: Union[str, list], with_stress=None, njobs=None, language=None) -> Union[str, list]:
        if not isinstance(text, (str, list)):
            raise TypeError(f'Phonemizer input must be list or str, not {type(text)}')
        try:
            from phonemizer.phonemize import phonemize
        except ImportError as e:
            raise RuntimeError('text input needs the `phonemizer` package and the espeak binary (not installed here): '
                               'pass a phoneme string to model.text_pipeline.tokenizer, or token ids to '
                               'predict(ids, encode=False)') from e
        many = isinstance(text, list)
        raw = phonemize([t.replace('-', self._EM_DASH) for t in (text if many else [text])],
                        language=language or self.language, backend='espeak', strip=True, preserve_punctuation=True,
                        with_stress=with_stress or self.with_stress, punctuation_marks=self._MARKS,
                        njobs=njobs or self.njobs, language_switch='remove-flags')
        out = [self._clean(r) for r in raw]
        return out if many else out[0]


class TextToTokens:
    """`model.text_pipeline` (reference data/text/__init__.py:7-21): text -> phonemes -> ids.  `phonemizer` is any
    callable text -> phoneme string; `default` attaches the espeak adapter above."""

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
        if phonemizer is None:
            phonemizer = Phonemizer(language=language, with_stress=with_stress, njobs=njobs)
        return cls(phonemizer, Tokenizer(add_start_end=add_start_end, model_breathing=model_breathing))
