"""Dataloader factory (mirrors bert4rec/dataloaders/__init__.py:39-60)."""
import abc

from .dataloader_utils import *  # noqa: F401,F403
from . import dataloader_utils, preprocessors, samplers  # noqa: F401
from .base_dataloader import BaseDataloader
from .bert4rec_dataloader import (BERT4RecBeautyDataloader, BERT4RecDataloader, BERT4RecML1MDataloader,
                                  BERT4RecML20MDataloader, BERT4RecRedditDataloader, BERT4RecSteamDataloader)


class BaseDataloaderFactory(abc.ABC):
    @abc.abstractmethod
    def create_ml_1m_dataloader(self, **kwargs) -> BaseDataloader:
        pass


class BERT4RecDataloaderFactory(BaseDataloaderFactory):
    def create_ml_1m_dataloader(self, **kwargs) -> BERT4RecDataloader:
        return BERT4RecML1MDataloader(**kwargs)

    def create_ml_20m_dataloader(self, **kwargs) -> BERT4RecDataloader:
        return BERT4RecML20MDataloader(**kwargs)

    def create_beauty_dataloader(self, **kwargs) -> BERT4RecDataloader:
        return BERT4RecBeautyDataloader(**kwargs)

    def create_steam_dataloader(self, **kwargs) -> BERT4RecDataloader:
        return BERT4RecSteamDataloader(**kwargs)

    def create_reddit_dataloader(self, **kwargs) -> BERT4RecDataloader:
        return BERT4RecRedditDataloader(**kwargs)


def get_dataloader_factory(identifier: str = "bert4rec") -> BaseDataloaderFactory:
    if identifier == "bert4rec":
        return BERT4RecDataloaderFactory()
    raise ValueError(f"{identifier} is not a known model/identifier!")
