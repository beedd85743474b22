"""Evaluator factory (mirrors bert4rec/evaluation/__init__.py:11-23)."""
from .base_evaluator import BaseEvaluator
from .bert4rec_evaluator import BERT4RecEvaluator, default_metrics  # noqa: F401
from .evaluation_metrics import *  # noqa: F401,F403

evaluators_map = {"bert4rec": BERT4RecEvaluator}


def get(identifier: str = "bert4rec", **kwargs) -> BaseEvaluator:
    if identifier in evaluators_map:
        return evaluators_map[identifier](**kwargs)
    raise ValueError(f"{identifier} is not known!")
