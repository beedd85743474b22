"""Sampler factory (mirrors bert4rec/dataloaders/samplers/__init__.py:17-31)."""
from typing import Union

from .base_sampler import BaseSampler
from .popular_random_sampler import PopularRandomSampler
from .popular_sampler import PopularSampler
from .random_sampler import RandomSampler

samplers_map = {"random": RandomSampler, "popular": PopularSampler, "pop_random": PopularRandomSampler,
                "popular_random": PopularRandomSampler}


def get(identifier: Union[str, BaseSampler] = "popular", **kwargs) -> BaseSampler:
    if isinstance(identifier, str) and identifier in samplers_map:
        return samplers_map[identifier](**kwargs)
    if isinstance(identifier, BaseSampler):
        return identifier
    raise ValueError(f"{identifier} is not known!")
