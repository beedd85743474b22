"""mirrors bert4rec/dataloaders/samplers/popular_sampler.py:53-71 (the `sample_size` most frequent items)."""
from .base_sampler import BaseSampler
from .. import dataloader_utils


class PopularSampler(BaseSampler):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None):
        super().__init__(source, vocab, sample_size)
        if self.source is not None:
            self.source = dataloader_utils.rank_items_by_popularity(self.source)

    def is_fully_prepared(self) -> bool:
        return self.source is not None and self.sample_size is not None

    def _get_parameters(self, source=None, vocab=None, sample_size=None):
        source, vocab, sample_size = super()._get_parameters(source, vocab, sample_size)
        if source is None:
            raise ValueError("The source argument has to be provided to the popular sampler but None was given.")
        return source, vocab, sample_size

    def sample(self, sample_size=None, source=None, vocab=None, without=None) -> list:
        source, vocab, sample_size = self._get_parameters(source, vocab, sample_size)
        _source = source.copy()
        if without is not None:
            wo = set(without)
            _source = [i for i in _source if i not in wo]
        if self.source is None:
            _source = dataloader_utils.rank_items_by_popularity(_source)
        return _source[:sample_size]

    def set_source(self, source: list):
        super().set_source(source)
        self.source = dataloader_utils.rank_items_by_popularity(self.source)
