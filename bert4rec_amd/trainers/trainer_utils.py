"""Loss and metric objects of the training surface (mirror bert4rec/trainers/trainer_utils.py).

On the GPU the arithmetic of both is fused into ONE HIP kernel pass over the logits (b4r_softmax_ce: per-row
log-sum-exp, picked logit, argmax), which BERT4RecModel.train_step / test_step call directly.  The callables below
exist so that user code written against the reference (`loss(y_true, y_pred)`, `masked_accuracy(y_true, y_pred)`) keeps
working on device tensors; they run the same kernel on a copy of the logits."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib


def _ce_sums(y_true: torch.Tensor, y_pred: torch.Tensor):
    if y_pred.device.type != "cuda":
        raise _lib.B4RError("bert4rec_amd computes only on the GPU: logits must be a device tensor")
    lib = _lib.load()
    V = y_pred.shape[-1]
    M = y_pred.numel() // V
    ld = (V + 3) // 4 * 4
    buf = torch.empty((M, ld), dtype=torch.float32, device=y_pred.device)
    buf[:, :V].copy_(y_pred.reshape(M, V))
    y = y_true.reshape(M).to(device=y_pred.device, dtype=torch.int64).contiguous()
    rows = torch.empty(4 * M, dtype=torch.float32, device=y_pred.device)
    state = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=y_pred.device)
    stream = torch.cuda.current_stream(y_pred.device).cuda_stream
    _lib.check(lib.b4r_softmax_ce(buf.data_ptr(), M, V, ld, y.data_ptr(), rows.data_ptr(), state.data_ptr(), 0, stream),
               "b4r_softmax_ce")
    return state.view(torch.float32)


class MaskedSparseCategoricalCrossentropy:
    """trainer_utils.py:4-23: sum(ce * (y_true != pad)) / sum(y_true != pad), a batch-global mean."""

    def __init__(self, pad_token: int = 0, reduction: str = "auto", name: str = None):
        if pad_token != 0:
            raise NotImplementedError("the fused loss kernel ignores slots with y_true == 0 (the reference's pad token)")
        self.pad_token = pad_token
        self.reduction = reduction
        self.name = name or "masked_sparse_categorical_crossentropy"

    def __call__(self, y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
        f = _ce_sums(y_true, y_pred)
        return f[_lib.ST_LOSS_SUM] / f[_lib.ST_VALID]

    call = __call__


def masked_accuracy(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """trainer_utils.py:49-60"""
    f = _ce_sums(y_true, y_pred)
    return f[_lib.ST_CORRECT_MASKED] / f[_lib.ST_VALID]


def sparse_categorical_accuracy(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """tf.keras.metrics.SparseCategoricalAccuracy on one batch (bert4rec_trainer.py:28-33)."""
    f = _ce_sums(y_true, y_pred)
    return f[_lib.ST_CORRECT_ALL] / f[_lib.ST_SLOTS_ALL]


class MaskedAccuracyMetric:
    """trainer_utils.py:26-46 (unused by the reference's trainer; kept for API parity)."""

    def __init__(self, pad_token: int = 0):
        self.pad_token = pad_token
        self.total = None

    def update_state(self, y_true, y_pred, sample_weight=None):
        self.total = masked_accuracy(y_true, y_pred)

    def result(self):
        return self.total
