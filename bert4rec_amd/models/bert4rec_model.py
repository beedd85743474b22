"""BERT4RecModel: the Keras-model surface of bert4rec/models/bert4rec_model.py:27-240 on the HIP engine.

  model(inputs, training=)   -> dict(sequence_output, pooled_output, encoder_outputs, mlm_logits)   :110-149
  model.compile / fit / train_step / test_step                                                      :151-192
  model.rank_items(encoder_input, items)                                                            :203-240

The arithmetic runs in libb4r_hip.so (include/b4r.h); this class only sequences calls and keeps the Keras bookkeeping
(metric names and averaging rules, History, callbacks)."""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from ..trainers import trainer_utils
from ..trainers.optimizers import AdamWeightDecay
from ..trainers import optimizers as _optimizers
from .components import networks

SPECIAL_TOKEN_IDS = [0, 1, 2]  # [PAD], [MASK], [UNK]: bert4rec_dataloader.py:38-43


class History:
    """Minimal stand-in for the object Keras fit() returns (bert4rec_trainer.py:62-68)."""

    def __init__(self):
        self.history: Dict[str, List[float]] = {}
        self.epoch: List[int] = []

    def _append(self, epoch: int, logs: Dict[str, float]):
        self.epoch.append(epoch)
        for k, v in logs.items():
            self.history.setdefault(k, []).append(v)


class _MetricLog:
    """Per-step copies of the 64-byte device state; read back once (no host sync inside the step loop)."""

    def __init__(self, device, capacity: int = 1024):
        self.device = device
        self.buf = torch.zeros((capacity, _lib.STATE_WORDS), dtype=torch.int32, device=device)
        self.n = 0
        self.batch_sizes: List[int] = []

    def reset(self):
        self.n = 0
        self.batch_sizes = []

    def push(self, state: torch.Tensor, batch_size: Optional[int]):
        """batch_size None (data-parallel rounds): the weight of the round in the epoch's loss is the all-reduced slot count of the
        state (rows of the JOINED batch x P) -- the same on every rank, also on one that had no batch in the round"""
        if self.n == self.buf.shape[0]:
            bigger = torch.zeros((2 * self.buf.shape[0], _lib.STATE_WORDS), dtype=torch.int32, device=self.device)
            bigger[: self.n].copy_(self.buf[: self.n])
            self.buf = bigger
        self.buf[self.n].copy_(state, non_blocking=True)
        self.n += 1
        self.batch_sizes.append(-1 if batch_size is None else batch_size)

    def results(self, prefix: str = "") -> Dict[str, float]:
        """Keras averaging rules: `loss` = batch-size-weighted mean of the per-batch losses; SparseCategoricalAccuracy =
        matches / slots over the epoch; masked_accuracy (a plain function metric) = unweighted mean of per-batch values."""
        if self.n == 0:
            return {}
        f = self.buf[: self.n].cpu().view(torch.float32).numpy().astype(np.float64)
        bs = np.asarray(self.batch_sizes, dtype=np.float64)
        bs = np.where(bs < 0, f[:, _lib.ST_SLOTS_ALL], bs)
        loss_b = f[:, _lib.ST_LOSS_SUM] / f[:, _lib.ST_VALID]
        macc_b = f[:, _lib.ST_CORRECT_MASKED] / f[:, _lib.ST_VALID]
        return {prefix + "loss": float((loss_b * bs).sum() / bs.sum()),
                prefix + "sparse_categorical_accuracy": float(f[:, _lib.ST_CORRECT_ALL].sum() / f[:, _lib.ST_SLOTS_ALL].sum()),
                prefix + "masked_accuracy": float(macc_b.mean())}


class BERT4RecModel:
    def __init__(self, encoder: networks.Bert4RecEncoder, customized_masked_lm: Any = None, mlm_activation="gelu",
                 mlm_initializer="glorot_uniform", name: str = "bert4rec",
                 special_token_ids: Optional[List[int]] = SPECIAL_TOKEN_IDS, **kwargs):
        if customized_masked_lm is not None:
            raise NotImplementedError("customized_masked_lm is not supported: the masked-LM head is a fused HIP path")
        if mlm_activation != "gelu":
            raise NotImplementedError("only mlm_activation='gelu' is implemented")
        self._config = {"encoder": encoder, "customized_masked_lm": customized_masked_lm,
                        "mlm_activation": mlm_activation, "mlm_initializer": mlm_initializer, "name": name}
        self.name = name
        self.encoder = encoder
        self.engine = encoder.engine
        self.device = encoder.device
        self.vocab_size = encoder.get_config()["vocab_size"]
        # the reference builds a -inf prediction mask for the special tokens and then disables it
        # (bert4rec_model.py:89-102): PAD/MASK/UNK logits are NOT suppressed.
        self.prediction_mask = None
        self.special_token_ids = special_token_ids
        self.inputs = ["input_word_ids", "input_mask", "masked_lm_positions"]
        self.optimizer: Optional[AdamWeightDecay] = None
        self.loss = None
        self.compiled_loss = None
        self.compiled_metrics = None
        self.metrics_names = ["loss", "sparse_categorical_accuracy", "masked_accuracy"]
        self.stop_training = False
        self._hp = None
        self._train_log = _MetricLog(self.device) if self.device.type == "cuda" else None
        self._eval_log = _MetricLog(self.device) if self.device.type == "cuda" else None
        self._trained_steps = 0

    @property
    def identifier(self):
        return "bert4rec"

    # ---- forward ------------------------------------------------------------------------------------------------------
    def __call__(self, inputs, training=None, mask=None) -> Dict[str, Any]:
        if isinstance(inputs, (list, tuple)):
            inputs = dict(zip(self.inputs, inputs))
        cb, keep = self.engine.prepare_batch(inputs)
        self.engine.forward(cb, training=bool(training), pooler=True)
        return self._outputs(cb)

    call = __call__

    def _outputs(self, cb, copy: bool = True) -> Dict[str, Any]:
        """bert4rec_model.py:139-149.  copy=True (the public call): tensors of their own, as the reference returns; the engine's
        workspace is overwritten by the next forward / train_step / test_step of the same batch shape."""
        out = self.encoder._outputs(cb, copy)
        if cb.P > 0:
            B, L, P = cb.B, cb.L, cb.P
            # [B,P,V] view into the (row-padded) logits buffer: values as in the reference, strides differ
            logits = self.engine.region("mlm_logits", B, L, P)
            view = torch.as_strided(logits, (B, P, self.vocab_size), (P * logits.stride(0), logits.stride(0), 1),
                                    logits.storage_offset())
            out["mlm_logits"] = view.contiguous() if copy else view
        return out

    # ---- compile / steps ------------------------------------------------------------------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None):
        """bert4rec_trainer.py:13-35.  Only the reference's own loss/metric set is fused on the device."""
        optimizer = _optimizers.get(optimizer if optimizer is not None else "adamw")
        if loss is None:
            loss = trainer_utils.MaskedSparseCategoricalCrossentropy()
        if not isinstance(loss, trainer_utils.MaskedSparseCategoricalCrossentropy):
            raise NotImplementedError("only MaskedSparseCategoricalCrossentropy is implemented as a fused device loss")
        self.optimizer = optimizer
        self.loss = self.compiled_loss = loss
        self.compiled_metrics = metrics if metrics is not None else ["sparse_categorical_accuracy", trainer_utils.masked_accuracy]
        self._hp = optimizer.kernel_config(self.engine.table, self.engine.n_params, self.engine.device)

    def _require_compiled(self):
        if self._hp is None:
            raise RuntimeError("The model needs to be compiled first (trainers.get(model=model).initialize_model()).")

    def _enqueue_train_step(self, cb, group=None):
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()
                                 and torch.distributed.get_world_size() > 1):
            self.engine.dp_train_step(self._hp, cb, group)
        else:
            self.engine.train_step(self._hp, cb)
        self._trained_steps += 1
        self.optimizer.iterations += 1

    def train_step(self, inputs) -> Dict[str, float]:
        """bert4rec_model.py:151-173.  Returns the running epoch metrics like Keras does."""
        self._require_compiled()
        cb, keep = self.engine.prepare_batch(inputs)
        if cb.P == 0 or keep.get("masked_lm_ids") is None:
            raise ValueError("train_step needs masked_lm_positions and masked_lm_ids")
        self._enqueue_train_step(cb)
        self._train_log.push(self.engine.state, cb.B)
        return self._train_log.results()

    def _enqueue_test_step(self, cb):
        # loss + the two accuracies only: the logits-free head where it exists (no [B*P, V] tensor is written)
        fused = self.engine.fused_head_supported()
        self.engine.begin_step()
        self.engine.forward(cb, training=False, pooler=False, fused_head=fused)
        self.engine.loss(cb, want_grad=False, fused_head=fused)

    def test_step(self, inputs) -> Dict[str, float]:
        """bert4rec_model.py:175-192"""
        cb, keep = self.engine.prepare_batch(inputs)
        if cb.P == 0 or keep.get("masked_lm_ids") is None:
            raise ValueError("test_step needs masked_lm_positions and masked_lm_ids")
        self._enqueue_test_step(cb)
        self._eval_log.push(self.engine.state, cb.B)
        return self._eval_log.results()

    def reset_metrics(self):
        self._train_log.reset()
        self._eval_log.reset()

    def evaluate(self, x: Iterable, steps: Optional[int] = None, prefix: str = "") -> Dict[str, float]:
        self._eval_log.reset()
        for i, batch in enumerate(x):
            if steps is not None and i >= steps:
                break
            cb, keep = self.engine.prepare_batch(batch)
            self._enqueue_test_step(cb)
            self._eval_log.push(self.engine.state, cb.B)
        return self._eval_log.results(prefix)

    def fit(self, x: Iterable, validation_data: Optional[Iterable] = None, epochs: int = 1, callbacks: Sequence = (),
            steps_per_epoch: Optional[int] = None, validation_steps: Optional[int] = None, verbose: int = 1) -> History:
        """Keras fit() as the reference drives it (bert4rec_trainer.py:62-68): per epoch all batches of `x`, then the
        validation pass; callbacks see the Keras metric names (val_masked_accuracy, ...)."""
        self._require_compiled()
        for ds in (x, validation_data):
            if ds is not None and hasattr(ds, "cache_on_device"):
                ds.cache_on_device(self.device)
        history = History()
        self.stop_training = False
        for cb_ in callbacks:
            if hasattr(cb_, "set_model"):
                cb_.set_model(self)
        rank, world = _dp_rank_world()
        if world > 1:   # data parallel: the ranks take the batches of an epoch in turn (same number each), with their own dropout masks
            if not hasattr(x, "__len__"):
                raise ValueError("data-parallel fit() needs a sized training set (every rank must take the same number of steps)")
            if not getattr(self, "_dp_seeded", False):
                self.engine.set_seed(self.engine.read_state()["seed"] + rank)
                self._dp_seeded = True
        for epoch in range(epochs):
            self._train_log.reset()
            for i, batch in enumerate(dp_shard(x, rank, world)):
                if steps_per_epoch is not None and i >= steps_per_epoch:
                    break
                if batch is None:      # the last round of the epoch has no batch for this rank: zero contribution, same update
                    self.engine.dp_idle_step(self._hp)
                    self._trained_steps += 1
                    self.optimizer.iterations += 1
                    self._train_log.push(self.engine.state, None)
                    continue
                cb, keep = self.engine.prepare_batch(batch)
                self._enqueue_train_step(cb)
                self._train_log.push(self.engine.state, cb.B if world == 1 else None)
            logs = self._train_log.results()
            if validation_data is not None:
                logs.update(self.evaluate(validation_data, validation_steps, prefix="val_"))
            history._append(epoch, logs)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()), flush=True)
            for cb_ in callbacks:
                if hasattr(cb_, "on_epoch_end"):
                    cb_.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb_ in callbacks:
            if hasattr(cb_, "on_train_end"):
                cb_.on_train_end()
        return history

    # ---- ranking ----------------------------------------------------------------------------------------------------------
    def _ranked_slot_hidden(self, encoder_input: Dict[str, torch.Tensor], slots: Optional[torch.Tensor] = None,
                            rows: Optional[torch.Tensor] = None):
        """Encoder forward (no logits, no head on the slots nobody ranks), then tfm MaskedLM's transform on the R slots with
        masked_lm_weights == 1 only (all slots when the key is absent).  The reference computes all [B, P, V] logits and
        keeps the valid slots afterwards (bert4rec_model.py:215-220).  Returns (hidden [R,H], slot index [R] = b*P+p,
        valid slots per batch row).  slots: the caller already has the slot indices (the evaluator, from sampling the candidates):
        nothing is read back to the host here, and the per-row counts are not formed (None)."""
        cb, keep = self.engine.prepare_batch(encoder_input)
        if cb.P == 0:
            raise ValueError("rank_items needs masked_lm_positions")
        B, L, P = cb.B, cb.L, cb.P
        # rows given: the caller vouches that the ranked slots are exactly the slots with masked_lm_ids != 0 (the evaluator checks it
        # once per resident batch) -- the last layer's feed-forward half then runs on those rows only
        enc = self.engine.encoder_forward(cb, training=False, ranked_rows_only=rows is not None and "masked_lm_ids" in keep)
        counts = None
        if slots is None:
            if "masked_lm_weights" in encoder_input and encoder_input["masked_lm_weights"] is not None:
                w = torch.as_tensor(encoder_input["masked_lm_weights"]).to(self.device).reshape(B, P) != 0
            else:
                w = torch.ones((B, P), dtype=torch.bool, device=self.device)
            slots = torch.nonzero(w.reshape(-1), as_tuple=False).reshape(-1)  # row-major => batch order, then slot order
            counts = w.sum(dim=1).tolist()
        if slots.numel() == 0:
            return None, slots, counts
        if rows is None:                                                           # (the evaluator keeps them for resident batches)
            pos = keep["masked_lm_positions"].reshape(-1)[slots].clamp(0, L - 1)   # tfm MaskedLM gathers position + b*L
            rows = torch.div(slots, P, rounding_mode="floor") * L + pos
        seq = self.engine.region("sequence_output", B, L, enc.P, encoder_only=enc.P > 0)   # (P > 0: the ranked-rows forward's own buffer)
        return self.engine.mlm_transform_rows(seq, rows), slots, counts

    def rank_items_tensor(self, encoder_input: Dict[str, torch.Tensor], candidates: Optional[torch.Tensor] = None,
                          ground_truth: Optional[torch.Tensor] = None, want_ranking: bool = True,
                          slots: Optional[torch.Tensor] = None, rows: Optional[torch.Tensor] = None):
        """Device-side core of rank_items: b4r_rank_candidates on every slot with masked_lm_weights == 1.  candidates:
        [R, C] int64 or None (whole vocabulary, ranked without materialising an [R, V] candidate list).
        Returns (ranking [R,C] int64, gt_rank [R] int32 or None, slot_index [R] int64 (b*P+p), rows_per_batch_entry (None when the
        caller passed `slots`))."""
        hidden, slots, counts = self._ranked_slot_hidden(encoder_input, slots, rows)
        R = int(slots.numel())
        if R == 0:
            return None, None, slots, counts
        if candidates is not None:
            candidates = torch.as_tensor(candidates).to(device=self.device, dtype=torch.int64).contiguous()
            if candidates.shape[0] != R:
                raise ValueError(f"{candidates.shape[0]} candidate lists for {R} masked slots")
        ranking, gt_rank, _ = self.engine.rank_candidates(hidden, None, candidates, ground_truth, want_ranking=want_ranking,
                                                          n_candidates=self.vocab_size, n_rows=R)
        return ranking, gt_rank, slots, counts

    def rank_items(self, encoder_input: dict, items: list = None):
        """bert4rec_model.py:203-240.  `items`: per batch row a list (one entry per masked slot) of candidate-id lists;
        returns per batch row a list of 1-D tensors: the candidates sorted by descending logit (ties: lower index first)."""
        cand = None
        if items is not None and len(items) > 0 and type(items[0]) is list:
            flat = [c for row in items for c in row]
            lens = {len(c) for c in flat}
            if len(lens) > 1:
                return self._rank_items_ragged(encoder_input, items)
            cand = torch.tensor(flat, dtype=torch.int64)
        ranking, _, slots, counts = self.rank_items_tensor(encoder_input, cand)
        out, r = [], 0
        for n in counts:
            out.append([ranking[r + j] for j in range(n)])
            r += n
        return out

    def _rank_items_ragged(self, encoder_input, items):
        """Candidate lists of different lengths: one kernel call per distinct length."""
        flat = [c for row in items for c in row]
        hidden, slots, counts = self._ranked_slot_hidden(encoder_input)
        if len(flat) != int(slots.numel()):
            raise ValueError(f"{len(flat)} candidate lists for {int(slots.numel())} masked slots")
        results: List[Optional[torch.Tensor]] = [None] * len(flat)
        for n in sorted({len(c) for c in flat}):
            idx = [i for i, c in enumerate(flat) if len(c) == n]
            cand = torch.tensor([flat[i] for i in idx], dtype=torch.int64)
            ranking, _, _ = self.engine.rank_candidates(hidden, torch.tensor(idx, dtype=torch.int64), cand, None)
            for j, i in enumerate(idx):
                results[i] = ranking[j]
        out, r = [], 0
        for n in counts:
            out.append(results[r:r + n])
            r += n
        return out

    # ---- weights ----------------------------------------------------------------------------------------------------------
    @property
    def trainable_variables(self) -> List[str]:
        return [e.name for e in self.engine.table]

    def get_weights(self) -> Dict[str, torch.Tensor]:
        """Variables under the reference's Keras names and shapes (CPU tensors)."""
        return self.engine.export_named()

    def set_weights(self, weights: Dict[str, torch.Tensor]) -> None:
        self.engine.load_named(weights)

    def save_weights(self, filepath) -> None:
        from safetensors.torch import save_file
        w = {k: v.contiguous() for k, v in self.get_weights().items()}
        save_file(w, str(filepath), metadata={"format": "bert4rec_amd", "model": self.name})

    def load_weights(self, filepath) -> None:
        """Weights only: like the reference's resume path the optimizer state is NOT restored (bert4rec_trainer.py:53-58)."""
        from safetensors.torch import load_file
        self.set_weights(load_file(str(filepath)))

    def get_config(self):
        return dict(self._config)

    @classmethod
    def from_config(cls, config, custom_object=None):
        return cls(**config)


def _dp_rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:   # pragma: no cover
        pass
    return 0, 1


def dp_shard(batches, rank: int, world: int):
    """The batches rank `rank` of `world` trains on in one epoch, one entry per ROUND (= per all-reduce): batch
    round * world + rank, or None where the last round has no batch left for this rank.  Every rank sees the same number of
    rounds, and EVERY batch is consumed by exactly one rank: a rank without a batch joins the round's all-reduce with zero
    gradients and zero sums (Engine.dp_idle_step), which is the reference's single-process step on the batches that are there
    (trainer_utils.py:19-22 normalises by the count of valid slots).  Round 3 dropped the trailing len % world batches -- with the
    reference's default reshuffle_each_iteration=False (dataloader_utils.py:306-311) the SAME batches were then never trained on.
    world == 1: all of them."""
    if world <= 1:
        yield from batches
        return
    n = len(batches)
    mine = iter(b for i, b in enumerate(batches) if i % world == rank)
    for r in range(-(-n // world)):
        yield next(mine) if r * world + rank < n else None
