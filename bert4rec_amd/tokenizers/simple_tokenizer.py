"""SimpleTokenizer: item string -> dense int id in insertion order (mirrors bert4rec/tokenizers/simple_tokenizer.py:119-138);
vocab file = `key|id` lines with the OS line separator (:104-116)."""
import numbers
import os
import pathlib
from collections.abc import Iterable

import numpy as np
import pandas as pd

from .base_tokenizer import BaseTokenizer


class SimpleTokenizer(BaseTokenizer):
    def __init__(self, vocab_file_path: pathlib.Path = None, extensible: bool = True):
        self._vocab = dict()
        self._delimiter = "|"
        self._reverse = None
        super().__init__(vocab_file_path=vocab_file_path, extensible=extensible)
        if self._vocab is None:
            self._vocab = dict()

    @property
    def identifier(self):
        return "simple"

    def clear_vocab(self):
        self._vocab = dict()
        self._vocab_size = 0
        self._reverse = None

    def tokenize(self, input, progress_bar: bool = False):
        if isinstance(input, bytes):
            input = input.decode()
        if isinstance(input, str):
            return self._tokenize_string(input)
        if isinstance(input, pd.Series):
            return input.map(self.tokenize)
        if isinstance(input, np.ndarray):
            input = input.tolist()
        if isinstance(input, Iterable):
            return [self.tokenize(v) for v in input]
        raise ValueError("The provided argument is not of a supported type")

    def detokenize(self, token, drop_tokens=None, progress_bar: bool = False):
        if isinstance(token, numbers.Number):
            return self._detokenize_token(int(token), drop_tokens)
        if isinstance(token, pd.Series):
            return token.map(lambda t: self.detokenize(t, drop_tokens))
        if hasattr(token, "tolist") and not isinstance(token, (list, tuple)):
            token = token.tolist()
            if isinstance(token, numbers.Number):
                return self._detokenize_token(int(token), drop_tokens)
        if isinstance(token, Iterable):
            values = []
            for t in token:
                v = self.detokenize(t, drop_tokens)
                if v is not None:
                    values.append(v)
            return values
        raise ValueError("The provided argument is not of a supported type")

    def import_vocab_from_file(self, vocab_file: pathlib.Path) -> bool:
        vocab_file = pathlib.Path(vocab_file)
        if not vocab_file.is_file():
            raise RuntimeError(f"The vocab file does not exist (yet) or is not located at {vocab_file}.")
        self.clear_vocab()
        with open(vocab_file, "rb") as file:
            lines = file.readlines()
        if len(lines) <= 0:
            raise ValueError(f"The given vocab file ({vocab_file}) is empty.")
        first = lines[0].decode()
        if self._delimiter not in first:
            raise ValueError(f"The given vocab file ({vocab_file}) does not contain \"{self._delimiter}\"-separated values.")
        if len(first.split(self._delimiter)) != 2:
            raise ValueError(f"The given vocab file ({vocab_file}) should contain \"{self._delimiter}\"-separated "
                             f"key-value-pairs per individual line.")
        for line in lines:
            parts = line.decode().split(self._delimiter)
            self._vocab[parts[0]] = int(parts[1])
        self._vocab_size = len(self._vocab)
        return True

    def export_vocab_to_file(self, file_path: pathlib.Path) -> bool:
        if len(self._vocab) <= 0:
            raise ValueError("The vocab of the tokenizer is empty and therefore can't be written to a file.")
        with open(file_path, "wb") as file:
            for key, token in self._vocab.items():
                file.write(bytes(key + self._delimiter + str(token) + os.linesep, "utf-8"))
        return True

    def _tokenize_string(self, string: str) -> int:
        tok = self._vocab.get(string)
        if tok is None:
            if not self._extensible:
                raise RuntimeError(f"\"{string}\" is not known!")
            tok = self._vocab_size
            self._vocab[string] = tok
            self._vocab_size += 1
            self._reverse = None
        return tok

    def _detokenize_token(self, token: int, drop_tokens=None):
        if self._reverse is None:
            self._reverse = {v: k for k, v in self._vocab.items()}
        value = self._reverse.get(token)
        if drop_tokens and value in drop_tokens:
            value = None
        return value
