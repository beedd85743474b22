"""The n most frequent items that are not banned (bert4rec/dataloaders/samplers/popular_sampler.py:53-71).  Deterministic:
the source is ranked by frequency once (dataloader_utils.rank_items_by_popularity) and every sample is a prefix of that
ranking after the banned items are dropped."""
from __future__ import annotations

from .. import dataloader_utils
from .base_sampler import BaseSampler


class PopularSampler(BaseSampler):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None):
        super().__init__(source, vocab, sample_size)
        self._ranked = self.source is not None   # self.source then holds the popularity ranking, not the raw log
        if self._ranked:
            self.source = dataloader_utils.rank_items_by_popularity(self.source)

    def is_fully_prepared(self) -> bool:
        return self.source is not None and self.sample_size is not None

    def _get_parameters(self, source: list = None, vocab: list = None, sample_size: int = None):
        source, vocab, n = super()._get_parameters(source, vocab, sample_size)
        if source is None:
            self._pick("source", None, required=True)
        return source, vocab, n

    def sample(self, sample_size: int = None, source: list = None, vocab: list = None, without: list = None) -> list:
        source, _, n = self._get_parameters(source, vocab, sample_size)
        banned = self._banned(without)
        ranking = [item for item in source if item not in banned]
        if not self._ranked:                      # raw log handed in at call time: rank what is left of it
            ranking = dataloader_utils.rank_items_by_popularity(ranking)
        return ranking[:n]

    def set_source(self, source: list):
        self.source = dataloader_utils.rank_items_by_popularity(list(source))
        self._ranked = True
