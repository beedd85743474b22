"""The two Keras callbacks the reference's scripts use: ModelCheckpoint (bert4rec_trainer.py:46-52) and EarlyStopping
(examples/bert4rec_ml_1m_example.py:70)."""
import pathlib

import numpy as np


class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_train_end(self):
        pass


def _mode(monitor: str, mode: str):
    if mode == "auto":
        mode = "max" if ("acc" in monitor or monitor.startswith("fmeasure")) else "min"
    if mode not in ("min", "max"):
        raise ValueError(f"mode must be auto, min or max, got {mode}")
    return mode


class ModelCheckpoint(Callback):
    """Weights-only checkpoint; save_best_only keeps the best `monitor` value (max for accuracies, like Keras 'auto')."""

    def __init__(self, filepath, monitor: str = "val_loss", save_best_only: bool = False, save_weights_only: bool = True,
                 mode: str = "auto"):
        super().__init__()
        if not save_weights_only:
            raise NotImplementedError("only weights-only checkpoints exist (the reference uses save_weights_only=True)")
        self.filepath = pathlib.Path(filepath)
        self.monitor, self.save_best_only = monitor, save_best_only
        self.mode = _mode(monitor, mode)
        self.best = -np.inf if self.mode == "max" else np.inf

    def weights_file(self) -> pathlib.Path:
        return self.filepath if self.filepath.suffix == ".safetensors" else self.filepath.with_suffix(".safetensors")

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        cur = logs.get(self.monitor)
        if self.save_best_only:
            if cur is None:
                return
            better = cur > self.best if self.mode == "max" else cur < self.best
            if not better:
                return
            self.best = cur
        # data-parallel runs: every rank holds the same weights and sees the same logs; rank 0 writes, the others wait for the file
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not multi or dist.get_rank() == 0:
            f = self.weights_file()
            f.parent.mkdir(parents=True, exist_ok=True)
            self.model.save_weights(f)
        if multi:
            dist.barrier()


class EarlyStopping(Callback):
    def __init__(self, monitor: str = "val_loss", min_delta: float = 0.0, patience: int = 0, verbose: int = 0, mode: str = "auto",
                 baseline=None, restore_best_weights: bool = False):
        """Keras' argument order (the reference passes {"monitor", "patience", "verbose"}: bert4rec_ml_1m_example.py:26-30)"""
        super().__init__()
        if baseline is not None:
            raise NotImplementedError("EarlyStopping(baseline=...) is not implemented (no reference script uses it)")
        self.verbose = verbose
        self.monitor, self.min_delta, self.patience = monitor, abs(min_delta), patience
        self.mode = _mode(monitor, mode)
        self.restore_best_weights = restore_best_weights
        self.best = -np.inf if self.mode == "max" else np.inf
        self.wait = 0
        self.best_weights = None
        self.stopped_epoch = None

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        better = cur - self.min_delta > self.best if self.mode == "max" else cur + self.min_delta < self.best
        if better:
            self.best, self.wait = cur, 0
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
        else:
            self.wait += 1
            if self.wait >= max(self.patience, 1):
                self.stopped_epoch = epoch
                self.model.stop_training = True
                if self.verbose:
                    print(f"Epoch {epoch + 1}: early stopping")
                if self.restore_best_weights and self.best_weights is not None:
                    self.model.set_weights(self.best_weights)
