"""What every evaluator offers: a metric set that is reset / read / saved together, a negative sampler, an `evaluate`.

Public names as in bert4rec/evaluation/base_evaluator.py:14-79; the results file is `eval_results.json` when a directory is given."""
import abc
import json
import logging
import pathlib
from typing import Union

from ..dataloaders import samplers
from .evaluation_metrics import EvaluationMetric  # noqa: F401  (re-exported for type annotations of subclasses)

RESULTS_FILE = "eval_results.json"
_log = logging.getLogger(__name__)


class BaseEvaluator(abc.ABC):
    def __init__(self, metrics: list, sampler: Union[str, "samplers.BaseSampler"] = "popular", dataloader=None):
        self.sampler = samplers.get(sampler)
        missing = [what for what in ("sample_size", "source") if getattr(self.sampler, what, None) is None]
        if missing:   # legal: evaluate() may receive them later
            _log.info("%s: sampler without %s", type(self).__name__, " and ".join(missing))
        self.dataloader = dataloader
        self._metrics = metrics
        self.reset_metrics()

    @abc.abstractmethod
    def evaluate(self, model, test_data) -> list:
        """rank the held-out item of every user of test_data against its negatives; returns the updated metric objects"""

    # ---- the metric set ------------------------------------------------------------------------------------------------------------
    def get_metrics(self) -> list:
        return self._metrics

    def reset_metrics(self) -> None:
        for m in self._metrics:
            m.reset()

    def get_metrics_results(self) -> dict:
        return {m.name: m.result() for m in self._metrics}

    def save_results(self, save_path: pathlib.Path) -> pathlib.Path:
        target = pathlib.Path(save_path)
        if target.is_dir():
            target = target / RESULTS_FILE
        target.write_text(json.dumps(self.get_metrics_results(), indent=4))
        return target
