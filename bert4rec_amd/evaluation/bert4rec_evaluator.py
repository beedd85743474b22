"""BERT4RecEvaluator (mirrors bert4rec/evaluation/bert4rec_evaluator.py:24-120): per masked slot draw 100 negatives
with the sampler (excluding the user's items and the ground truth), append the ground truth as candidate 100, rank the 101
candidates and feed the 1-based rank of the ground truth to every metric.

The ranking itself is one b4r_rank_candidates launch per batch (scores + stable ordering + rank lookup on the GPU)
instead of the reference's per-user python loop of tf.gather / tf.argsort calls, the metric sums of a batch are one
b4r_rank_metrics launch into device accumulators that are read back ONCE per evaluate() (no host synchronisation per batch),
and under torch.distributed the batches are dealt round-robin to the ranks with one all-reduce of the sums at the end
("shard users across ranks ... one 8-float all-reduce").  With the popularity sampler the 100
negatives of every slot of a batch are drawn by one b4r_sample_candidates launch as well (`device_sampling`, default on
when the model runs on a GPU): same distribution as the reference's per-slot np.random.choice, own random stream;
`sample_candidates` keeps the reference's host procedure (pinned by the golden vectors)."""
from typing import Union

import numpy as np
import torch

from ..dataloaders import samplers
from .base_evaluator import BaseEvaluator
from .evaluation_metrics import HR, MAP, NDCG, Counter, EvaluationMetric, gain_table


def default_metrics():
    """bert4rec_evaluator.py:12-21"""
    return [Counter(name="Valid Ranks"), NDCG(1), NDCG(5), NDCG(10), HR(1), HR(5), HR(10), MAP()]


class BERT4RecEvaluator(BaseEvaluator):
    def __init__(self, metrics: list = None, sampler: Union[str, "samplers.BaseSampler"] = "pop_random", dataloader=None,
                 device_sampling: bool = True, seed: int = 0):
        self.device_sampling = device_sampling
        self._dev = None   # (engine, float64 gain sums [n_metrics], int64 user count [1]) on the GPU
        self._seed = int(seed)
        self._draws = 0
        self._logp = None
        self._short = None   # device flag: some row had fewer drawable items than the sample size (checked once per evaluation)
        self._slots = None
        self._rows = None
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

    def evaluate(self, model, test_data, group=None) -> list:
        """bert4rec_evaluator.py:46-58.  With an initialised torch.distributed process group (`group`, default WORLD) rank k
        evaluates batches k, k + world, ... and every rank ends with the metrics of ALL users."""
        if self.dataloader is None and not self.sampler.is_fully_prepared():
            raise ValueError("The evaluator has to be either initialized with a dataloader or a fully prepared sampler "
                             "has to be given.")
        rank, world = _dist_rank_world(group)
        before = [m.partial() for m in self._metrics]
        for i, batch in enumerate(test_data):
            if i % world == rank:
                self.evaluate_batch(model, batch)
        # one rank's "too few drawable items" must not strand the others in the all-reduce (every rank draws from its own stream and
        # sees its own batches, so the ranks can disagree): with world > 1 the flag travels IN that collective and every rank raises
        # after it -- an input error stays an error, it does not become a hang
        short = self._flush_device_sums(raise_on_short=(world == 1))
        if world > 1:
            short = self._merge_across_ranks(before, group, short)
            if short:
                raise ValueError(self._short_message())
        return self._metrics

    # ---- device-side accumulation ---------------------------------------------------------------------------------------
    def _device_sums(self, engine):
        if self._dev is None or self._dev[0] is not engine:
            import torch as _t
            n = len(self._metrics)
            self._dev = (engine, _t.zeros(n, dtype=_t.float64, device=engine.device), _t.zeros(1, dtype=_t.int64, device=engine.device))
        return self._dev

    def _short_message(self) -> str:
        return (f"The exclusion lists reduce the vocab too much to take a sample of size {self.sampler.sample_size} "
                f"(since no duplicates are allowed).")

    def _flush_device_sums(self, raise_on_short: bool = True) -> bool:
        """one device -> host copy for the whole evaluation: fold the accumulated sums into the metric objects.  Returns True (or
        raises, raise_on_short) when the sampler kernel flagged a row with fewer drawable items than the sample size: nothing of
        that evaluation is then folded in."""
        if self._dev is None:
            return False
        _, sums, users = self._dev
        if getattr(self, "_short", None) is not None and bool(self._short.cpu()[0]):   # checked once, not per batch
            self._short.zero_()
            sums.zero_()    # nothing of the aborted evaluation may leak into the next one
            users.zero_()
            if raise_on_short:
                raise ValueError(self._short_message())
            return True
        sums_h, users_h = sums.cpu().tolist(), int(users.cpu()[0])
        for m, g in zip(self._metrics, sums_h):
            m.absorb(g, users_h)
        sums.zero_()
        users.zero_()
        return False

    def _merge_across_ranks(self, before, group, short: bool = False) -> bool:
        """all-reduce what THIS evaluate() call added on each rank: [gain sums | user count | short flag] as float64.  Returns True
        when ANY rank reported a short row; the metrics of every rank are then back at their state before the call."""
        import torch.distributed as dist
        mine = [(m.partial()[0] - b[0], m.partial()[1] - b[1]) for m, b in zip(self._metrics, before)]
        # the buffer's device follows the BACKEND (a rank that saw no batch has no device accumulators, but must still join an
        # nccl collective with a device tensor)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
        buf = torch.tensor([g for g, _ in mine] + [float(mine[0][1]) if mine else 0.0, 1.0 if short else 0.0],
                           dtype=torch.float64, device=dev)
        dist.all_reduce(buf, group=group)
        tot = buf.cpu().tolist()
        if tot[-1] > 0.0:                                     # some rank's sampler came up short: the same decision on every rank
            for m, b in zip(self._metrics, before):
                m.restore(b[0], b[1])
            return True
        for m, b, g_all in zip(self._metrics, before, tot[:-2]):
            m.restore(b[0] + g_all, b[1] + int(round(tot[-2])))   # the same two additions on every rank: identical metrics everywhere
        return False

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
        # a sampler seeded the reference's way (np.random.seed(seed) per call, popular_random_sampler.py:88-90) keeps ITS stream: the
        # device sampler draws from another one, so it only replaces unseeded samplers
        return (self.device_sampling and isinstance(self.sampler, samplers.PopularRandomSampler)
                and getattr(self.sampler, "seed", None) is None
                and not self.sampler.allow_duplicates and self.sampler.is_fully_prepared()
                and getattr(model, "engine", None) is not None and model.engine.device.type == "cuda"
                and model.engine.cfg.vocab_size * 4 <= 150 * 1024   # b4r_sample_candidates keeps the V keys in LDS
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
        pre = getattr(test_batch, "slot_index", None)   # dataloader_utils.ResidentBatch: a batch that stays in HBM
        resident = (pre is not None and torch.is_tensor(pre) and pre.device.type == dev.type and pre.ndim == 2 and pre.shape[1] == 2
                    and (dev.index is None or pre.device.index == dev.index))
        cached = test_batch.eval_cache if resident else None
        if cached is not None:                                  # a resident batch seen before
            self._slots, gt, exclude, self._rows = cached
        else:
            w = torch.as_tensor(test_batch["masked_lm_weights"]).to(dev)
            ids_t = torch.as_tensor(test_batch["masked_lm_ids"]).to(dev)
            labels = torch.as_tensor(test_batch["labels"]).to(dev)
            if resident:
                b_idx, p_idx = pre[:, 0], pre[:, 1]             # found on the host copy: no read-back
            else:
                b_idx, p_idx = torch.nonzero(w != 0, as_tuple=True)   # row-major: batch order, then slot order (one read-back per batch)
            self._slots = b_idx * w.shape[1] + p_idx            # handed to rank_items_tensor: it need not look for them again
            self._rows = None
            if b_idx.numel() == 0:
                return torch.empty((0, self.sampler.sample_size + 1), dtype=torch.int64), torch.empty((0,), dtype=torch.int64)
            gt = ids_t[b_idx, p_idx].to(torch.int64)
            exclude = labels[b_idx].to(torch.int64)             # the user's whole sequence (bert4rec_evaluator.py:86-95)
            if resident:                                        # per-batch constants of a resident batch are formed once
                L = int(torch.as_tensor(test_batch["input_word_ids"]).shape[1])
                pos = torch.as_tensor(test_batch["masked_lm_positions"]).to(dev)[b_idx, p_idx].clamp(0, L - 1)
                rows = b_idx * L + pos                          # tfm MaskedLM gathers position + b*L
                # handing `rows` to the model lets it run the last layer's feed-forward half on the rows of the slots with
                # masked_lm_ids != 0 only: allowed when those ARE the ranked slots (checked here, once per resident batch)
                same = bool(((ids_t != 0) == (w != 0)).all()) and int(torch.unique(rows).numel()) == int(rows.numel())
                self._rows = rows if same else None
                test_batch.eval_cache = (self._slots, gt, exclude, self._rows)
        if self._slots.numel() == 0:
            return torch.empty((0, self.sampler.sample_size + 1), dtype=torch.int64), torch.empty((0,), dtype=torch.int64)
        self._draws += 1
        if self._short is None or self._short.device != dev:
            self._short = torch.zeros(1, dtype=torch.bool, device=dev)
        rank, _ = _dist_rank_world(None)   # every data-parallel rank draws from its own stream
        cand = eng.sample_candidates(self._logp, exclude, gt, self.sampler.sample_size,
                                     seed=((self._seed << 32) ^ self._draws) ^ (rank << 48), short_flag=self._short)
        return cand, gt

    def evaluate_batch(self, model, test_batch: dict, candidates=None, ground_truth=None):
        """bert4rec_evaluator.py:60-120 for one batch.  Returns the ground-truth ranks: a device int32 tensor when the metric
        sums are accumulated on the GPU (flushed by evaluate() / get_metrics_results()), else a numpy array."""
        slots = None
        if candidates is None:
            if self._device_sampler_ready(model):
                candidates, ground_truth = self.sample_candidates_device(model, test_batch)
                slots = self._slots
            else:
                candidates, ground_truth = self.sample_candidates(test_batch)
        if len(candidates) == 0:
            return []
        extra = {} if slots is None else {"slots": slots}   # (models without the shortcut keep working)
        if slots is not None and getattr(self, "_rows", None) is not None:
            extra["rows"] = self._rows
        _, gt_rank, _, _ = model.rank_items_tensor(test_batch, torch.as_tensor(candidates), torch.as_tensor(ground_truth),
                                                   want_ranking=False, **extra)
        engine = getattr(model, "engine", None)
        if engine is not None and gt_rank.is_cuda and len(self._metrics) <= 32:
            _, sums, users = self._device_sums(engine)
            table = gain_table(self._metrics)
            engine.rank_metrics(gt_rank, [f for f, _ in table], [k for _, k in table], sums, users)
            return gt_rank
        ranks = gt_rank.cpu().numpy().astype(np.int64)
        for metric in self._metrics:
            metric.update(ranks)
        return ranks

    def get_metrics(self) -> list:
        self._flush_device_sums()
        return self._metrics

    def get_metrics_results(self) -> dict:
        self._flush_device_sums()
        return super().get_metrics_results()

    def reset_metrics(self) -> None:
        if getattr(self, "_dev", None) is not None:
            self._dev[1].zero_()
            self._dev[2].zero_()
        if getattr(self, "_short", None) is not None:
            self._short.zero_()
        super().reset_metrics()


def _dist_rank_world(group):
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(group), dist.get_world_size(group)
    except Exception:   # pragma: no cover
        pass
    return 0, 1
