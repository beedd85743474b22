"""Interface of a tokenizer: items <-> integer ids, with a vocabulary that can be frozen, exported and re-imported.

Public names as in bert4rec/tokenizers/base_tokenizer.py."""
import abc
import pathlib


class BaseTokenizer(abc.ABC):
    def __init__(self, vocab_file_path: pathlib.Path = None, extensible: bool = True):
        self._max_len = 1
        self._vocab_size = 0
        self._vocab = None
        self._extensible = extensible
        if vocab_file_path is not None and pathlib.Path(vocab_file_path).is_file():
            self._extensible = False
            self.import_vocab_from_file(pathlib.Path(vocab_file_path))

    @property
    @abc.abstractmethod
    def identifier(self):
        pass

    def generate_vocab_from_ds(self, ds):
        """grow the vocabulary over an iterable of item sequences (a dataset of 1-D vectors in the reference)"""
        for items in ds:
            self.tokenize(items)

    @property
    def max_seq_len(self) -> int:
        return self._max_len

    def get_vocab(self):
        return self._vocab

    def get_vocab_size(self) -> int:
        return self._vocab_size

    def enable_extensibility(self):
        self._extensible = True

    def disable_extensibility(self):
        self._extensible = False

    @abc.abstractmethod
    def clear_vocab(self):
        pass

    @abc.abstractmethod
    def tokenize(self, input, progress_bar: bool = False):
        pass

    @abc.abstractmethod
    def detokenize(self, token, drop_tokens=None, progress_bar: bool = False):
        pass

    @abc.abstractmethod
    def import_vocab_from_file(self, vocab_file: pathlib.Path) -> bool:
        pass

    @abc.abstractmethod
    def export_vocab_to_file(self, file_path: pathlib.Path) -> bool:
        pass
