"""What every trainer offers: compile the model, run epochs, validate, collect callbacks, stamp a model wrapper after a run.

Public names and arguments as in bert4rec/trainers/base_trainer.py:9-55."""
import abc
import pathlib
from datetime import datetime


class BaseTrainer(abc.ABC):
    def __init__(self, model):
        self.model = model
        self.optimizer = self.loss = None   # set by initialize_model
        self.metrics, self.callbacks = [], []

    # ---- for the concrete trainer -----------------------------------------------------------------------------------------------
    @abc.abstractmethod
    def initialize_model(self, optimizer=None, loss=None, metrics: list = None):
        """compile the model with this optimizer / loss / metric set (each None: the trainer's default)"""

    @abc.abstractmethod
    def train(self, train_ds, val_ds, checkpoint_path: pathlib.Path = None, epochs: int = 50, steps_per_epoch: int = None,
              validation_steps: int = None):
        """run `epochs` epochs over train_ds, validating on val_ds; returns the history object of the fit"""

    @abc.abstractmethod
    def validate(self):
        """one pass over the validation set"""

    # ---- shared ----------------------------------------------------------------------------------------------------------------
    def append_callback(self, callback):
        self.callbacks.append(callback)

    def update_wrapper_meta_info(self, wrapper, dataloader):
        """after training: when, and on which dataset (ModelWrapper.update_meta)"""
        stamp = {"last_trained": str(datetime.now()), "trained_on_dataset": dataloader.dataset_identifier}
        wrapper.update_meta(stamp)
