"""Popularity-weighted negatives (bert4rec/dataloaders/samplers/popular_random_sampler.py:77-126): draw
``n + |banned|`` items with ``np.random.choice(vocab, size, replace, p)`` after ``np.random.seed(seed)``, drop the banned
ones, keep the first n -- the reference's numpy calls, so equal seeds give equal samples (golden vectors).

p[i] = occurrences of vocab[i] in the source / len(source), built with ONE counting pass (the reference calls
``source.count`` per vocabulary entry: O(V * N)); the floats are the same quotients.  ``probability_distribution`` stays a
public attribute: the evaluator's device sampler (b4r_sample_candidates) reads it."""
from __future__ import annotations

import collections

import numpy as np

from .base_sampler import BaseSampler


class PopularRandomSampler(BaseSampler):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None, allow_duplicates: bool = False,
                 seed: int = None):
        super().__init__(source, vocab, sample_size)
        self.vocab = vocab                        # shared with the caller, as the reference keeps it
        self.allow_duplicates = allow_duplicates
        self.seed = seed
        self.probability_distribution: list = []
        self._refresh()

    def _refresh(self) -> None:
        if self.source is not None and self.vocab is not None:
            self.probability_distribution = self._popularity(self.source, self.vocab)

    @staticmethod
    def _popularity(source: list, vocab: list) -> list:
        seen = collections.Counter(source)
        n = len(source)
        return [seen[item] / n if item in seen else 0 / n for item in vocab]

    def is_fully_prepared(self) -> bool:
        return (self.vocab is not None and self.sample_size is not None
                and len(self.probability_distribution) == len(self.vocab))

    def sample(self, sample_size: int = None, source: list = None, vocab: list = None, allow_duplicates: bool = None,
               seed: int = None, without: list = None) -> list:
        source, vocab, n = self._get_parameters(source, vocab, sample_size)
        np.random.seed(self.seed if seed is None else seed)
        source = self._pick("source", source, required=True)
        vocab = self._pick("vocab", vocab, required=True)
        repeat = self.allow_duplicates if allow_duplicates is None else allow_duplicates
        if not repeat:
            self._check_capacity(n, len(vocab), "a vocabulary")
        if not self.probability_distribution:
            self.probability_distribution = self._popularity(source, vocab)
        banned = self._banned(without)
        draws = n + len(banned)                   # enough to survive the removal of every banned item
        if not repeat:
            self._check_capacity(draws, len(vocab), f"a vocabulary that also has to absorb {len(banned)} banned items,")
        drawn = np.random.choice(vocab, draws, repeat, self.probability_distribution).tolist()
        return [item for item in drawn if item not in banned][:n]

    def set_source(self, source: list):
        super().set_source(source)
        self._refresh()

    def set_vocab(self, vocab: list):
        super().set_vocab(vocab)
        self._refresh()
