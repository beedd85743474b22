"""Administrative record carried next to a model (what it is, which tokenizer and dataset it was trained with, when).

Public names as in bert4rec/models/model_wrapper.py:6-53 (`model`, `get_meta_config`, `update_meta`, `delete_keys_from_meta`,
the `_custom_objects` class attribute that loaders consult)."""
from typing import Iterable, Union

META_FIELDS = ("tokenizer", "last_trained", "trained_on_dataset")   # known from the start, unset


class ModelWrapper:
    _custom_objects: dict = {}

    def __init__(self, model):
        self._model = model
        self._meta_config = dict.fromkeys(META_FIELDS)
        self._meta_config["model"] = model.name

    model = property(lambda self: self._model, doc="the wrapped model")

    def get_meta_config(self) -> dict:
        """the live record, not a copy: savers serialise exactly what update_meta / delete_keys_from_meta left"""
        return self._meta_config

    def update_meta(self, updated_info: dict) -> bool:
        self._meta_config.update(updated_info)
        return True

    def delete_keys_from_meta(self, keys: Union[Iterable[str], str]) -> bool:
        """a single key or several; unknown keys are ignored"""
        for key in ([keys] if isinstance(keys, str) else keys):
            self._meta_config.pop(key, None)
        return True
