"""BERT4RecEvaluator (mirrors bert4rec/evaluation/bert4rec_evaluator.py:24-120): per masked slot draw 100 negatives
with the sampler (excluding the user's items and the ground truth), append the ground truth as candidate 100, rank the 101
candidates and feed the 1-based rank of the ground truth to every metric.

The ranking itself is one b4r_rank_candidates launch per batch (scores + stable ordering + rank lookup on the GPU)
instead of the reference's per-user python loop of tf.gather / tf.argsort calls.  With the popularity sampler the 100
negatives of every slot of a batch are drawn by one b4r_sample_candidates launch as well (`device_sampling`, default on
when the model runs on a GPU): same distribution as the reference's per-slot np.random.choice, own random stream;
`sample_candidates` keeps the reference's host procedure (pinned by the golden vectors)."""
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
    def __init__(self, metrics: list = None, sampler: Union[str, "samplers.BaseSampler"] = "pop_random", dataloader=None,
                 device_sampling: bool = True, seed: int = 0):
        self.device_sampling = device_sampling
        self._seed = int(seed)
        self._draws = 0
        self._logp = None
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

    def _device_sampler_ready(self, model) -> bool:
        return (self.device_sampling and isinstance(self.sampler, samplers.PopularRandomSampler)
                and not self.sampler.allow_duplicates and self.sampler.is_fully_prepared()
                and getattr(model, "engine", None) is not None and model.engine.device.type == "cuda"
                and all(isinstance(t, (int, np.integer)) for t in self.sampler.vocab[:8]))

    def sample_candidates_device(self, model, test_batch: dict):
        """All slots of the batch in one launch.  The sampler's vocab entries are token ids and its
        probability_distribution is aligned with them (popular_random_sampler.py:66-69); ids missing from it have p = 0."""
        eng = model.engine
        if self._logp is None:
            V = eng.cfg.vocab_size
            p = np.zeros(V, dtype=np.float64)
            ids = np.asarray(self.sampler.vocab, dtype=np.int64)
            ok = (ids >= 0) & (ids < V)
            p[ids[ok]] = np.asarray(self.sampler.probability_distribution, dtype=np.float64)[ok]
            with np.errstate(divide="ignore"):
                self._logp = torch.from_numpy(np.log(p).astype(np.float32)).to(eng.device)
        dev = eng.device
        w = torch.as_tensor(test_batch["masked_lm_weights"]).to(dev) != 0
        ids_t = torch.as_tensor(test_batch["masked_lm_ids"]).to(dev)
        labels = torch.as_tensor(test_batch["labels"]).to(dev)
        b_idx, p_idx = torch.nonzero(w, as_tuple=True)          # row-major: batch order, then slot order
        if b_idx.numel() == 0:
            return torch.empty((0, self.sampler.sample_size + 1), dtype=torch.int64), torch.empty((0,), dtype=torch.int64)
        gt = ids_t[b_idx, p_idx].to(torch.int64)
        exclude = labels[b_idx].to(torch.int64)                 # the user's whole sequence (bert4rec_evaluator.py:86-95)
        self._draws += 1
        cand = eng.sample_candidates(self._logp, exclude, gt, self.sampler.sample_size,
                                     seed=(self._seed << 32) ^ self._draws)
        return cand, gt

    def evaluate_batch(self, model, test_batch: dict, candidates=None, ground_truth=None):
        if candidates is None:
            if self._device_sampler_ready(model):
                candidates, ground_truth = self.sample_candidates_device(model, test_batch)
            else:
                candidates, ground_truth = self.sample_candidates(test_batch)
        if len(candidates) == 0:
            return []
        _, gt_rank, _, _ = model.rank_items_tensor(test_batch, torch.as_tensor(candidates), torch.as_tensor(ground_truth))
        ranks = gt_rank.cpu().numpy().astype(np.int64)
        for rank in ranks.tolist():
            for metric in self._metrics:
                metric.update(rank)
        return ranks
