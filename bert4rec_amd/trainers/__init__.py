"""Trainer factory (mirrors bert4rec/trainers/__init__.py:10-23)."""
from . import optimizers, trainer_utils  # noqa: F401
from .base_trainer import BaseTrainer
from .bert4rec_trainer import BERT4RecTrainer
from .callbacks import Callback, EarlyStopping, ModelCheckpoint  # noqa: F401

trainers_map = {"bert4rec": BERT4RecTrainer}


def get(identifier: str = "bert4rec", **kwargs) -> BaseTrainer:
    if identifier in trainers_map:
        return trainers_map[identifier](**kwargs)
    raise ValueError(f"{identifier} is not known!")
