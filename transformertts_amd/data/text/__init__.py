"""Text front end mirror (reference data/text/__init__.py:7-21, data/text/tokenizer.py).

Only the Tokenizer is on the hot path's construction (ForwardTransformer.__init__ builds it and
its vocab size feeds the embedding, model/models.py:375-383).  The Phonemizer needs the
third-party `phonemizer` package + the espeak binary, which are outside the path (SURVEY.md section
2 row 9): it is imported lazily and raises a clear error when absent."""
from __future__ import annotations

import re
from typing import List, Union

from .symbols import _punctuations, all_phonemes


class Tokenizer:
    """char -> id with pad id 0 (reference data/text/tokenizer.py:9-47)."""

    def __init__(self, start_token='>', end_token='<', pad_token='/', add_start_end=True, alphabet=None,
                 model_breathing=True):
        self.alphabet = sorted(set(alphabet)) if alphabet else list(all_phonemes)
        self.idx_to_token = {i: s for i, s in enumerate(self.alphabet, start=1)}
        self.idx_to_token[0] = pad_token
        self.token_to_idx = {s: [i] for i, s in self.idx_to_token.items()}
        self.vocab_size = len(self.alphabet) + 1
        self.add_start_end = add_start_end
        if add_start_end:
            self.start_token_index = len(self.alphabet) + 1
            self.end_token_index = len(self.alphabet) + 2
            self.vocab_size += 2
            self.idx_to_token[self.start_token_index] = start_token
            self.idx_to_token[self.end_token_index] = end_token
        self.model_breathing = model_breathing
        if model_breathing:
            self.breathing_token_index = self.vocab_size
            self.token_to_idx[' '] = self.token_to_idx[' '] + [self.breathing_token_index]
            self.vocab_size += 1
            self.breathing_token = '@'
            self.idx_to_token[self.breathing_token_index] = self.breathing_token
            self.token_to_idx[self.breathing_token] = [self.breathing_token_index]

    def __call__(self, sentence: str) -> List[int]:
        seq = [i for c in sentence for i in self.token_to_idx[c]]   # unknown chars raise KeyError
        if self.model_breathing:
            seq = [self.breathing_token_index] + seq
        if self.add_start_end:
            seq = [self.start_token_index] + seq + [self.end_token_index]
        return seq

    def decode(self, sequence) -> str:
        return ''.join(self.idx_to_token[int(t)] for t in sequence)


class Phonemizer:
    """espeak front end (reference data/text/tokenizer.py:50-106); needs `phonemizer` + espeak."""

    def __init__(self, language: str, with_stress: bool, njobs=4):
        self.language, self.with_stress, self.njobs = language, with_stress, njobs
        self.special_hyphen = '—'
        self.punctuation = ';:,.!?¡¿—…"«»“”'
        self._ws = re.compile(r'\s+')
        self._ws_punct = re.compile(rf'\s*([{_punctuations}])\s*')

    def __call__(self, text: Union[str, list], with_stress=None, njobs=None, language=None):
        try:
            from phonemizer.phonemize import phonemize
        except ImportError as e:   # pragma: no cover - not installable offline
            raise RuntimeError('text -> phoneme conversion needs the `phonemizer` package and the espeak '
                               'binary; pass token ids with predict(..., encode=False) instead') from e
        one = isinstance(text, str)
        items = [text] if one else list(text)
        items = [t.replace('-', self.special_hyphen) for t in items]
        ph = phonemize(items, language=language or self.language, backend='espeak', strip=True,
                       preserve_punctuation=True, with_stress=with_stress or self.with_stress,
                       punctuation_marks=self.punctuation, njobs=njobs or self.njobs,
                       language_switch='remove-flags')
        out = []
        for t in ph:
            t = t.replace(self.special_hyphen, '-')
            t = ''.join(c for c in t if c in all_phonemes)
            t = re.sub(self._ws_punct, r'\1', re.sub(self._ws, ' ', t)).strip()
            out.append(t)
        return out[0] if one else out


class TextToTokens:
    def __init__(self, phonemizer: Phonemizer, tokenizer: Tokenizer):
        self.phonemizer, self.tokenizer = phonemizer, tokenizer

    def __call__(self, input_text):
        return self.tokenizer(self.phonemizer(input_text))

    @classmethod
    def default(cls, language: str, add_start_end: bool, with_stress: bool, model_breathing: bool, njobs=1):
        return cls(Phonemizer(language=language, njobs=njobs, with_stress=with_stress),
                   Tokenizer(add_start_end=add_start_end, model_breathing=model_breathing))
