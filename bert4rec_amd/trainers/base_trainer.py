"""mirrors bert4rec/trainers/base_trainer.py:9-55"""
import abc
import datetime
import pathlib


class BaseTrainer(abc.ABC):
    def __init__(self, model):
        self.model = model
        self.optimizer = None
        self.loss = None
        self.metrics = []
        self.callbacks = []

    @abc.abstractmethod
    def initialize_model(self, optimizer=None, loss=None, metrics: list = None):
        pass

    @abc.abstractmethod
    def train(self, train_ds, val_ds, checkpoint_path: pathlib.Path = None, epochs: int = 50,
              steps_per_epoch: int = None, validation_steps: int = None):
        pass

    def update_wrapper_meta_info(self, wrapper, dataloader):
        wrapper.update_meta({"last_trained": str(datetime.datetime.now()),
                             "trained_on_dataset": dataloader.dataset_identifier})

    @abc.abstractmethod
    def validate(self):
        pass

    def append_callback(self, callback):
        self.callbacks.append(callback)
