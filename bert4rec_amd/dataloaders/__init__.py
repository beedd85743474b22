"""Dataloaders and their factory.

Public names as in bert4rec/dataloaders/__init__.py:39-60 (`get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**kwargs)`
and its four siblings); here the factory is a table from dataset key to dataloader class, and the `create_<key>_dataloader`
methods are generated from that table."""
import abc
from typing import Dict, Type

from .dataloader_utils import *  # noqa: F401,F403
from . import dataloader_utils, preprocessors, samplers  # noqa: F401
from .base_dataloader import BaseDataloader
from .bert4rec_dataloader import (BERT4RecBeautyDataloader, BERT4RecDataloader, BERT4RecML1MDataloader,
                                  BERT4RecML20MDataloader, BERT4RecRedditDataloader, BERT4RecSteamDataloader)

DATASET_KEYS = ("ml_1m", "ml_20m", "beauty", "steam", "reddit")


class BaseDataloaderFactory(abc.ABC):
    """One `create_<key>_dataloader(**kwargs)` per key of DATASET_KEYS; a concrete factory supplies `table()`."""

    @abc.abstractmethod
    def table(self) -> Dict[str, Type[BaseDataloader]]:
        ...

    def create(self, key: str, **kwargs) -> BaseDataloader:
        classes = self.table()
        if key not in classes:
            raise ValueError(f"{type(self).__name__} has no dataloader for {key!r} (known: {', '.join(sorted(classes))})")
        return classes[key](**kwargs)


def _creator(key: str):
    def create(self, **kwargs):
        return self.create(key, **kwargs)
    create.__name__ = f"create_{key}_dataloader"
    create.__doc__ = f"the {key} dataloader of this factory; keyword arguments go to its constructor"
    return create


for _key in DATASET_KEYS:
    setattr(BaseDataloaderFactory, f"create_{_key}_dataloader", _creator(_key))


class BERT4RecDataloaderFactory(BaseDataloaderFactory):
    _TABLE = {"ml_1m": BERT4RecML1MDataloader, "ml_20m": BERT4RecML20MDataloader, "beauty": BERT4RecBeautyDataloader,
              "steam": BERT4RecSteamDataloader, "reddit": BERT4RecRedditDataloader}

    def table(self) -> Dict[str, Type[BaseDataloader]]:
        return self._TABLE


_FACTORIES = {"bert4rec": BERT4RecDataloaderFactory}


def get_dataloader_factory(identifier: str = "bert4rec") -> BaseDataloaderFactory:
    try:
        return _FACTORIES[identifier]()
    except KeyError:
        raise ValueError(f"{identifier} is not a known model/identifier!") from None
