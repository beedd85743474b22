"""Ranking-quality statistics over the 1-based rank of the ground-truth item.

Public surface = the names the reference's evaluator and examples use (bert4rec/evaluation/evaluation_metrics.py:47-112:
``Counter``, ``HitRatio``/``HR``, ``NormalizedDiscountedCumulativeGain``/``NDCG``, ``MeanAveragePrecision``/``MAP`` with
``name`` / ``update(rank)`` / ``result()`` / ``reset()``); the values are pinned by the reference's known-answer tests
(tests/evaluators_tests/evaluation_metrics_tests.py:28-104, reused as data in tests/golden/reference_goldens.json).

Own design: every statistic is "the sum over users of a gain g(rank), divided by the number of users" (or the bare number of
users).  So one object holds two numbers, ``update`` takes a single rank or a whole rank vector, and partial sums computed
elsewhere -- the GPU kernel b4r_rank_metrics over a batch's ranks, or another data-parallel rank's share of the users -- are
folded in with ``absorb``.  ``GAIN_*`` are the gain families the device kernel knows (include/b4r.h)."""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple, Union

import numpy as np

GAIN_COUNT, GAIN_HIT, GAIN_NDCG, GAIN_RECIPROCAL = 0, 1, 2, 3

RankLike = Union[int, np.integer, Sequence[int], np.ndarray]


class EvaluationMetric:
    """sum_u g(rank_u) and the number of users u seen so far."""

    family = GAIN_COUNT
    averaged = True   # result = gain sum / users; False: result = users

    def __init__(self, name: str, cutoff: int = 0):
        self.name = name
        self.cutoff = int(cutoff)
        self._gain_sum = 0.0
        self._users = 0

    # -- the statistic ---------------------------------------------------------------------------------------------
    def gain(self, ranks: np.ndarray) -> np.ndarray:
        """g(rank) element-wise (float64) for an int64 vector of 1-based ranks."""
        return np.zeros(ranks.shape, dtype=np.float64)

    # -- accumulation ----------------------------------------------------------------------------------------------
    def update(self, rank: RankLike):
        r = np.asarray(rank, dtype=np.int64).reshape(-1)
        if r.size:
            # left-to-right float64 accumulation, continuing from the running sum: the value a one-rank-at-a-time
            # caller (the reference's loop) would reach, bit for bit
            self._gain_sum = float(np.cumsum(np.concatenate(([self._gain_sum], self.gain(r))))[-1])
            self._users += int(r.size)
        return self.result()

    def absorb(self, gain_sum: float, users: int) -> None:
        """fold in a partial (gain sum, user count) accumulated elsewhere"""
        self._gain_sum += float(gain_sum)
        self._users += int(users)

    def partial(self) -> Tuple[float, int]:
        return self._gain_sum, self._users

    def restore(self, gain_sum: float, users: int) -> None:
        """set the accumulators (the data-parallel merge: state before the evaluation + the sum over all ranks, the same
        two numbers on every rank)"""
        self._gain_sum, self._users = float(gain_sum), int(users)

    def reset(self) -> None:
        self._gain_sum, self._users = 0.0, 0

    def result(self):
        if not self.averaged:
            return self._users
        return self._gain_sum / self._users if self._users else 0

    def __repr__(self):
        return f"{type(self).__name__}({self.name!r}: {self.result()})"


class Counter(EvaluationMetric):
    """number of ranks seen ("Valid Ranks" in the default metric set, bert4rec_evaluator.py:12-21)"""

    family, averaged = GAIN_COUNT, False

    def __init__(self, name: str = "Counter", initial_value: int = 0):
        super().__init__(name)
        self._base = int(initial_value)

    def result(self):
        return self._base + self._users


class _AtK(EvaluationMetric):
    def __init__(self, k: int, name: str, initial_value: int = 0):
        super().__init__(f"{name}@{k}", cutoff=k)
        self._k = int(k)


class HitRatio(_AtK):
    """share of users whose ground truth is ranked within the first k"""

    family = GAIN_HIT

    def __init__(self, k: int, name: str = "HitRatio", initial_value: int = 0):
        super().__init__(k, name, initial_value)

    def gain(self, ranks):
        return (ranks <= self._k).astype(np.float64)


class NormalizedDiscountedCumulativeGain(_AtK):
    """one relevant item per user: DCG = 1 / log2(rank + 1) inside the cut-off, ideal DCG = 1"""

    family = GAIN_NDCG

    def __init__(self, k: int, name: str = "NormalizedDiscountedCumulativeGain", initial_value: int = 0):
        super().__init__(k, name, initial_value)

    def gain(self, ranks):
        out = np.zeros(ranks.shape, dtype=np.float64)
        inside = ranks <= self._k
        # np.log2 of one integer at a time: a vectorised log2 may round the last bit differently from the scalar routine
        # the known-answer values were produced with
        out[inside] = [1.0 if r == 1 else 1 / np.log2(r + 1) for r in ranks[inside].tolist()]
        return out


class MeanAveragePrecision(EvaluationMetric):
    """one relevant item per user: average precision = 1 / rank"""

    family = GAIN_RECIPROCAL

    def __init__(self, name: str = "MeanAveragePrecision", initial_value: int = 0):
        super().__init__(name)

    def gain(self, ranks):
        return np.asarray([1 / r for r in ranks.tolist()], dtype=np.float64)


class HR(HitRatio):
    def __init__(self, k: int, name: str = "HR", initial_value: int = 0):
        super().__init__(k, name, initial_value)


class NDCG(NormalizedDiscountedCumulativeGain):
    def __init__(self, k: int, name: str = "NDCG", initial_value: int = 0):
        super().__init__(k, name, initial_value)


class MAP(MeanAveragePrecision):
    def __init__(self, name: str = "MAP", initial_value: int = 0):
        super().__init__(name, initial_value)


# the reference's intermediate base class name, kept importable
RatioEvaluationMetric = EvaluationMetric


def gain_table(metrics: Iterable[EvaluationMetric]) -> List[Tuple[int, int]]:
    """(family, cutoff) per metric: the description b4r_rank_metrics takes"""
    return [(m.family, m.cutoff) for m in metrics]


def update_all(metrics: Iterable[EvaluationMetric], ranks: RankLike) -> None:
    for m in metrics:
        m.update(ranks)
