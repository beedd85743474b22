"""mirrors bert4rec/evaluation/base_evaluator.py:14-79"""
import abc
import json
import pathlib
from typing import Union

from ..dataloaders import samplers
from .evaluation_metrics import EvaluationMetric


class BaseEvaluator(abc.ABC):
    def __init__(self, metrics: list, sampler: Union[str, "samplers.BaseSampler"] = "popular", dataloader=None):
        self.sampler = samplers.get(sampler)
        self._metrics = metrics
        self.dataloader = dataloader
        self.reset_metrics()

    def reset_metrics(self) -> None:
        for metric in self._metrics:
            metric.reset()

    @abc.abstractmethod
    def evaluate(self, model, test_data) -> list:
        pass

    def get_metrics(self) -> list:
        return self._metrics

    def get_metrics_results(self) -> dict:
        return {metric.name: metric.result() for metric in self._metrics}

    def save_results(self, save_path: pathlib.Path) -> pathlib.Path:
        save_path = pathlib.Path(save_path)
        if save_path.is_dir():
            save_path = save_path.joinpath("eval_results.json")
        with open(save_path, "w") as f:
            json.dump(self.get_metrics_results(), f, indent=4)
        return save_path
