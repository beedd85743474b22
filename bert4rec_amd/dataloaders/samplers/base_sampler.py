"""Negative samplers for the evaluator: common argument handling.

API surface of bert4rec/dataloaders/samplers/base_sampler.py (constructor keywords source / vocab / sample_size, the
setters, ``sample(sample_size, source, vocab, without)``, ``is_fully_prepared``); the body is this package's own: a sampler
is a ``_draw(pool arguments, n, banned)`` routine, and every call-time argument is resolved against the constructor's by
one helper."""
from __future__ import annotations

import abc
from typing import Iterable, List, Optional


class BaseSampler(abc.ABC):
    # what an argument is called in error messages
    _LABELS = {"source": "item source", "vocab": "vocabulary", "sample_size": "sample size"}

    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None):
        self._check_size(sample_size)
        self.source = None if source is None else list(source)
        self.vocab = None if vocab is None else list(vocab)
        self.sample_size = sample_size

    # ---- argument plumbing ------------------------------------------------------------------------------------------
    @staticmethod
    def _check_size(n: Optional[int]) -> None:
        if n is not None and n < 0:
            raise ValueError(f"sample_size must be >= 0, got {n}")

    def _pick(self, field: str, given, required: bool = False):
        """call-time value if there is one, else the constructor's; ValueError when a required one is missing"""
        value = getattr(self, field) if given is None else given
        if value is None and required:
            raise ValueError(f"{type(self).__name__} has no {self._LABELS[field]}: pass `{field}` to the constructor or to sample()")
        return value

    def _get_parameters(self, source: list = None, vocab: list = None, sample_size: int = None):
        n = self._pick("sample_size", sample_size, required=True)
        self._check_size(n)
        return self._pick("source", source), self._pick("vocab", vocab), n

    @staticmethod
    def _banned(without: Optional[Iterable]) -> frozenset:
        return frozenset(without) if without is not None else frozenset()

    @staticmethod
    def _check_capacity(n: int, available: int, what: str) -> None:
        if n > available:
            raise ValueError(f"cannot draw {n} distinct items from {what} of {available}")

    # ---- interface --------------------------------------------------------------------------------------------------
    @abc.abstractmethod
    def sample(self, sample_size: int = None, source: list = None, vocab: list = None, without: list = None) -> list:
        ...

    @abc.abstractmethod
    def is_fully_prepared(self) -> bool:   # True when sample() can run on the constructor's arguments alone
        ...

    # the setters copy, like the constructor: a caller's later edits of its list do not reach the sampler
    def set_sample_size(self, sample_size: int): self.sample_size = sample_size
    def set_vocab(self, vocab: list): self.vocab = list(vocab)
    def set_source(self, source: list): self.source = list(source)
