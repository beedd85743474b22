"""BERT4RecEvaluator (mirrors bert4rec/evaluation/bert4rec_evaluator.py:24-120): per masked slot draw 100 negatives
with the sampler (excluding the user's items and the ground truth), append the ground truth as candidate 100, rank the 101
candidates and feed the 1-based rank of the ground truth to every metric.

The ranking itself is one b4r_rank_candidates launch per batch (scores + stable ordering + rank lookup on the GPU)
instead of the reference's per-user python loop of tf.gather / tf.argsort calls."""
from typing import Union

import numpy as np
import torch

from ..dataloaders import samplers
from .base_evaluator import BaseEvaluator
from .evaluation_metrics import HR, MAP, NDCG, Counter, EvaluationMetric


def default_metrics():
    """bert4rec_evaluator.py:12-21"""
    return [Counter(name="Valid Ranks"), NDCG(1), NDCG(5), NDCG(10), HR(1), HR(5), HR(10), MAP()]


class BERT4RecEvaluator(BaseEvaluator):
    def __init__(self, metrics: list = None, sampler: Union[str, "samplers.BaseSampler"] = "pop_random", dataloader=None):
        if metrics is None:
            metrics = default_metrics()
        if isinstance(sampler, str):
            sampler_config = {"sample_size": 100}
            if dataloader is not None:
                vocab = dataloader.tokenizer.get_vocab()
                tokenized_vocab = dataloader.tokenizer.tokenize(vocab)
                sampler_config.update({"source": dataloader.create_item_list_tokenized(), "vocab": tokenized_vocab})
            sampler = samplers.get(sampler, **sampler_config)
        super().__init__(metrics, sampler, dataloader)

    def evaluate(self, model, test_data) -> list:
        if self.dataloader is None and not self.sampler.is_fully_prepared():
            raise ValueError("The evaluator has to be either initialized with a dataloader or a fully prepared sampler "
                             "has to be given.")
        for batch in test_data:
            self.evaluate_batch(model, batch)
        return self._metrics

    def sample_candidates(self, test_batch: dict):
        """bert4rec_evaluator.py:75-108 -> (candidates [R,101] int64, ground truth [R] int64), slots in batch order."""
        w = torch.as_tensor(test_batch["masked_lm_weights"]).cpu().numpy() != 0
        ids = torch.as_tensor(test_batch["masked_lm_ids"]).cpu().numpy()
        labels = torch.as_tensor(test_batch["labels"]).cpu().numpy()
        cands, gts = [], []
        for b in range(w.shape[0]):
            remove_base = labels[b].tolist()
            for p in np.nonzero(w[b])[0]:
                gt = int(ids[b, p])
                sampled = self.sampler.sample(without=remove_base + [gt])
                sampled.append(gt)   # ground truth is the LAST candidate (index 100)
                cands.append(sampled)
                gts.append(gt)
        return np.asarray(cands, dtype=np.int64), np.asarray(gts, dtype=np.int64)

    def evaluate_batch(self, model, test_batch: dict, candidates=None, ground_truth=None):
        if candidates is None:
            candidates, ground_truth = self.sample_candidates(test_batch)
        if len(candidates) == 0:
            return []
        _, gt_rank, _, _ = model.rank_items_tensor(test_batch, torch.as_tensor(candidates), torch.as_tensor(ground_truth))
        ranks = gt_rank.cpu().numpy().astype(np.int64)
        for rank in ranks.tolist():
            for metric in self._metrics:
                metric.update(rank)
        return ranks
