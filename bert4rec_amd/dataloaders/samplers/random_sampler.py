"""mirrors bert4rec/dataloaders/samplers/random_sampler.py:63-79 (uniform, np.random.choice after np.random.seed)."""
import numpy as np

from .base_sampler import BaseSampler


class RandomSampler(BaseSampler):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None, allow_duplicates: bool = False,
                 seed: int = None):
        super().__init__(source, vocab, sample_size)
        if self.vocab is None and self.source is not None:
            self.vocab = list(set(self.source))
        self.allow_duplicates = allow_duplicates
        self.seed = seed

    def is_fully_prepared(self) -> bool:
        return self.vocab is not None and self.sample_size is not None

    def _get_parameters(self, source=None, vocab=None, sample_size=None, allow_duplicates=None, seed=None):
        source, vocab, sample_size = super()._get_parameters(source, vocab, sample_size)
        if vocab is None and source is not None and self.source is None:
            vocab = list(set(source))
        if vocab is None:
            raise ValueError("No vocab or any other source has been given to the random sampler.")
        if seed is None:
            seed = self.seed
        np.random.seed(seed)
        if allow_duplicates is None:
            allow_duplicates = self.allow_duplicates
        if allow_duplicates is False and sample_size > len(vocab):
            raise ValueError("When no duplicates are allowed in the final sample then the sample size "
                             f"(given sample size: {sample_size})) can not be greater than the length of the vocab "
                             f"(length of the vocab: {len(vocab)})")
        return source, vocab, sample_size, allow_duplicates

    def sample(self, sample_size=None, source=None, vocab=None, allow_duplicates=None, seed=None, without=None) -> list:
        source, vocab, sample_size, allow_duplicates = self._get_parameters(source, vocab, sample_size, allow_duplicates, seed)
        _source = vocab.copy()
        if without is not None:
            wo = set(without)
            _source = [i for i in _source if i not in wo]
        return np.random.choice(_source, size=sample_size, replace=allow_duplicates).tolist()

    def set_source(self, source: list):
        if not self.allow_duplicates:
            source = list(set(source.copy()))
        super().set_source(source)
