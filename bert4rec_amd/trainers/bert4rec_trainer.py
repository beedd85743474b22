"""BERT4RecTrainer (mirrors bert4rec/trainers/bert4rec_trainer.py:9-71)."""
import pathlib

from . import optimizers, trainer_utils
from .base_trainer import BaseTrainer
from .callbacks import ModelCheckpoint


class BERT4RecTrainer(BaseTrainer):
    def __init__(self, model):
        super().__init__(model)

    def initialize_model(self, optimizer=None, loss=None, metrics: list = None):
        """bert4rec_trainer.py:13-35: AdamWeightDecay defaults, MaskedSparseCategoricalCrossentropy,
        [SparseCategoricalAccuracy, masked_accuracy]."""
        if optimizer is None:
            optimizer = optimizers.get("adamw")
        self.optimizer = optimizer
        if loss is None:
            loss = trainer_utils.MaskedSparseCategoricalCrossentropy()
        self.loss = loss
        if metrics is None:
            metrics = ["sparse_categorical_accuracy", trainer_utils.masked_accuracy]
        self.metrics = metrics
        self.model.compile(optimizer=optimizer, loss=loss, metrics=metrics)

    def train(self, train_ds, val_ds, checkpoint_path: pathlib.Path = None, epochs: int = 50,
              steps_per_epoch: int = None, validation_steps: int = None):
        """bert4rec_trainer.py:37-68: best-val_masked_accuracy weights-only checkpoint; resume = load the weights of an
        existing checkpoint (optimizer state is not restored, as in the reference :53-58)."""
        if checkpoint_path:
            checkpoint_path = pathlib.Path(checkpoint_path)
            cb = ModelCheckpoint(filepath=checkpoint_path, save_weights_only=True, monitor="val_masked_accuracy",
                                 save_best_only=True)
            self.append_callback(cb)
            if cb.weights_file().is_file():
                self.model.load_weights(cb.weights_file())
        return self.model.fit(x=train_ds, validation_data=val_ds, epochs=epochs, callbacks=self.callbacks,
                              steps_per_epoch=steps_per_epoch, validation_steps=validation_steps)

    def validate(self):
        pass
