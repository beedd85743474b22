"""mirrors bert4rec/dataloaders/samplers/popular_random_sampler.py:77-126: popularity-weighted np.random.choice.

Same draws as the reference for equal seeds.  The probability table is built with one Counter pass instead of the
reference's `source.count(item)` per vocabulary entry (O(V*N) python, :119-126); the resulting floats are identical
(count / total)."""
import collections

import numpy as np

from .base_sampler import BaseSampler


class PopularRandomSampler(BaseSampler):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None, allow_duplicates: bool = False,
                 seed: int = None):
        super().__init__(source, vocab, sample_size)
        self.vocab = vocab
        self.probability_distribution = []
        self.allow_duplicates = allow_duplicates
        self.seed = seed
        if self.source is not None and self.vocab is not None:
            self._determine_probability_distribution(self.source, self.vocab)

    def is_fully_prepared(self) -> bool:
        if self.vocab is None or self.sample_size is None:
            return False
        return len(self.vocab) == len(self.probability_distribution)

    def _get_parameters(self, source=None, vocab=None, sample_size=None, allow_duplicates=None, seed=None):
        source, vocab, sample_size = super()._get_parameters(source, vocab, sample_size)
        if seed is None:
            seed = self.seed
        np.random.seed(seed)
        if source is None:
            raise ValueError("The source argument has to be given either during the initialization of the sampler or as "
                             "an argument in the sample method call when working with the popular random sampler.")
        if vocab is None:
            raise ValueError("The vocab argument has to be given either during the initialization of the sampler or as "
                             "an argument in the sample method call when working with the popular random sampler.")
        if allow_duplicates is None:
            allow_duplicates = self.allow_duplicates
        if allow_duplicates is False and sample_size > len(vocab):
            raise ValueError("When no duplicates are allowed in the final sample then the sample size "
                             f"(given sample size: {sample_size})) can not be greater than the length of the vocab "
                             f"(length of the vocab: {len(vocab)})")
        return source, vocab, sample_size, allow_duplicates

    def sample(self, sample_size=None, source=None, vocab=None, allow_duplicates=None, seed=None, without=None) -> list:
        source, vocab, sample_size, allow_duplicates = self._get_parameters(source, vocab, sample_size, allow_duplicates, seed)
        if not self.probability_distribution:
            self._determine_probability_distribution(source, vocab)
        size = sample_size
        if without is not None:
            without = list(set(without))
            size += len(without)
        if not allow_duplicates and size > len(vocab):
            raise ValueError(f"The given without list (length: {len(without)} reduces the vocab (length: {len(vocab)}) "
                             f"too much to take a sample of size {sample_size} (since no duplicates are allowed).")
        sample = np.random.choice(vocab, size, allow_duplicates, self.probability_distribution).tolist()
        if without is not None:
            wo = set(without)
            sample = [i for i in sample if i not in wo]
        return sample[:sample_size]

    def _determine_probability_distribution(self, source: list, vocab: list):
        counts = collections.Counter(source)
        total_items = len(source)
        self.probability_distribution = [counts.get(item, 0) / total_items for item in vocab]

    def set_source(self, source: list):
        super().set_source(source)
        if self.vocab is not None:
            self._determine_probability_distribution(self.source, self.vocab)

    def set_vocab(self, vocab: list):
        super().set_vocab(vocab)
        if self.source is not None:
            self._determine_probability_distribution(self.source, self.vocab)
