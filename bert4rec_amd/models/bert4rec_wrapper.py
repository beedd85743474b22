"""BERT4RecModelWrapper: save / load of a model directory (mirrors bert4rec/models/bert4rec_wrapper.py:21-124).

On-disk layout: `model_weights.safetensors` (variables under the reference's Keras names and shapes, so a converter
from/to a Keras checkpoint is a rename-free tensor copy), `encoder_config.json`, `meta_config.json` (same keys as the
reference: model, tokenizer, last_trained, trained_on_dataset, encoder_config) and `vocab.txt` (`key|id` lines)."""
import json
import pathlib

from .bert4rec_model import BERT4RecModel
from .components import networks
from .model_wrapper import ModelWrapper
from . import model_utils as utils

_ENCODER_CONFIG_FILE_NAME = "encoder_config.json"
_META_CONFIG_FILE_NAME = "meta_config.json"
_TOKENIZER_VOCAB_FILE_NAME = "vocab.txt"
_MODEL_WEIGHTS_FILE_NAME = "model_weights.safetensors"

_JSON_KEYS = ("vocab_size", "hidden_size", "num_layers", "num_attention_heads", "max_sequence_length", "inner_dim",
              "output_dropout", "attention_dropout")


class BERT4RecModelWrapper(ModelWrapper):
    def __init__(self, model: BERT4RecModel):
        super().__init__(model)
        self.update_meta({"model": "BERT4Rec", "encoder_config": self._encoder_json()})

    def _encoder_json(self) -> dict:
        cfg = self.model.encoder.get_config()
        return {k: cfg[k] for k in _JSON_KEYS}

    def save(self, save_path: pathlib.Path, tokenizer=None, mode: int = 0) -> bool:
        save_path = utils.determine_model_path(pathlib.Path(save_path), mode)
        if self.model.compiled_loss is None:
            raise RuntimeError("The model can't be saved without a loss. The model needs to be compiled first.")
        if self.model._trained_steps == 0:
            # bert4rec_wrapper.py:63-68: Keras refuses to save until one train_step has built the compiled metrics
            raise RuntimeError("The model can't be saved yet, as it is not fully instantiated (run one train_step first).")
        save_path.mkdir(parents=True, exist_ok=True)
        self.model.save_weights(save_path.joinpath(_MODEL_WEIGHTS_FILE_NAME))
        with open(save_path.joinpath(_ENCODER_CONFIG_FILE_NAME), "w") as f:
            json.dump(self._encoder_json(), f, indent=4)
        if tokenizer:
            tokenizer.export_vocab_to_file(save_path.joinpath(_TOKENIZER_VOCAB_FILE_NAME))
            self.update_meta({"tokenizer": tokenizer.identifier})
        with open(save_path.joinpath(_META_CONFIG_FILE_NAME), "w") as f:
            json.dump(self._meta_config, f, indent=4)
        return True

    @classmethod
    def load(cls, save_path: pathlib.Path, mode: int = 0, device=None) -> dict:
        """Returns {"model_wrapper": wrapper, "tokenizer": tokenizer or None} like bert4rec_wrapper.py:85-124."""
        from .. import tokenizers
        save_path = utils.determine_model_path(pathlib.Path(save_path), mode)
        with open(save_path.joinpath(_ENCODER_CONFIG_FILE_NAME)) as f:
            enc_cfg = json.load(f)
        encoder = networks.Bert4RecEncoder(device=device, **enc_cfg)
        model = BERT4RecModel(encoder)
        model.load_weights(save_path.joinpath(_MODEL_WEIGHTS_FILE_NAME))
        wrapper = cls(model)
        meta_path = save_path.joinpath(_META_CONFIG_FILE_NAME)
        tokenizer = None
        if meta_path.is_file():
            with open(meta_path) as f:
                wrapper.update_meta(json.load(f))
        vocab_path = save_path.joinpath(_TOKENIZER_VOCAB_FILE_NAME)
        if wrapper.get_meta_config().get("tokenizer") and vocab_path.is_file():
            tokenizer = tokenizers.get(wrapper.get_meta_config()["tokenizer"])
            tokenizer.import_vocab_from_file(vocab_path)
        return {"model_wrapper": wrapper, "tokenizer": tokenizer}
