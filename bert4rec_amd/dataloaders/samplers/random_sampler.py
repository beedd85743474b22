"""Uniform negatives (bert4rec/dataloaders/samplers/random_sampler.py:63-79): the banned items are removed from the pool
first, then ONE ``np.random.choice`` call after ``np.random.seed(seed)`` -- the same numpy calls as the reference, so equal
seeds give equal samples (pinned by tests/golden/reference_goldens.json)."""
from __future__ import annotations

import numpy as np

from .base_sampler import BaseSampler


def _distinct(items) -> list:
    return list(set(items))


class RandomSampler(BaseSampler):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None, allow_duplicates: bool = False,
                 seed: int = None):
        super().__init__(source, vocab, sample_size)
        self.allow_duplicates = allow_duplicates
        self.seed = seed
        if self.vocab is None and self.source is not None:
            self.vocab = _distinct(self.source)   # no vocabulary given: whatever occurs in the source

    def is_fully_prepared(self) -> bool:
        return None not in (self.vocab, self.sample_size)

    def _pool(self, source, vocab) -> list:
        if vocab is None and source is not None and self.source is None:
            vocab = _distinct(source)             # a call-time source on a sampler that never had one
        if vocab is None:
            raise ValueError("RandomSampler needs a vocabulary (or a source to derive it from)")
        return vocab

    def sample(self, sample_size: int = None, source: list = None, vocab: list = None, allow_duplicates: bool = None,
               seed: int = None, without: list = None) -> list:
        source, vocab, n = self._get_parameters(source, vocab, sample_size)
        pool = self._pool(source, vocab)
        np.random.seed(self.seed if seed is None else seed)
        repeat = self.allow_duplicates if allow_duplicates is None else allow_duplicates
        if not repeat:
            self._check_capacity(n, len(pool), "a vocabulary")
        banned = self._banned(without)
        if banned:
            pool = [item for item in pool if item not in banned]
        return np.random.choice(pool, size=n, replace=repeat).tolist()

    def set_source(self, source: list):
        super().set_source(source if self.allow_duplicates else _distinct(source))
