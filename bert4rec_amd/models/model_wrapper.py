"""mirrors bert4rec/models/model_wrapper.py:6-53"""
from typing import Union


class ModelWrapper:
    _custom_objects: dict = {}

    def __init__(self, model):
        self._model = model
        self._meta_config = {"model": model.name, "tokenizer": None, "last_trained": None, "trained_on_dataset": None}

    @property
    def model(self):
        return self._model

    def get_meta_config(self) -> dict:
        return self._meta_config

    def update_meta(self, updated_info: dict) -> bool:
        self._meta_config.update(updated_info)
        return True

    def delete_keys_from_meta(self, keys: Union[list, str]) -> bool:
        if isinstance(keys, str):
            keys = [keys]
        for key in keys:
            self._meta_config.pop(key, None)
        return True
