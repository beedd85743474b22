"""Bert4RecEncoder: same constructor surface as bert4rec/models/components/networks/bert4rec_encoder.py:62-80, computing
on the HIP engine (embedding gather + position add + LayerNorm, N post-LN transformer blocks, tanh pooler)."""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from ...engine import Engine, make_model_config


def _default_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class Bert4RecEncoder:
    def __init__(self, vocab_size: int, hidden_size: int = 768, num_layers: int = 12, num_attention_heads: int = 12,
                 max_sequence_length: int = 512, inner_dim: int = 3072, inner_activation: Any = "gelu",
                 output_dropout: float = 0.1, attention_dropout: float = 0.1, initializer: Any = None,
                 output_range: Optional[int] = None, embedding_width: Optional[int] = None,
                 embedding_layer: Any = None, norm_first: bool = False, with_dense_inputs: bool = False,
                 device=None, seed: int = 3, name: str = "bert4rec_encoder", **kwargs):
        # legacy kwargs of the V1 encoder (bert4rec_encoder.py:82-93)
        kwargs.pop("dict_outputs", None)
        kwargs.pop("return_all_encoder_outputs", None)
        if "intermediate_size" in kwargs:
            inner_dim = kwargs.pop("intermediate_size")
        if "activation" in kwargs:
            inner_activation = kwargs.pop("activation")
        if "dropout_rate" in kwargs:
            output_dropout = kwargs.pop("dropout_rate")
        if "attention_dropout_rate" in kwargs:
            attention_dropout = kwargs.pop("attention_dropout_rate")
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")
        if embedding_width is None:
            embedding_width = hidden_size
        # options of the reference constructor that no shipped config uses and the kernels do not implement
        if inner_activation not in ("gelu", None):
            raise NotImplementedError("only inner_activation='gelu' (erf form) is implemented")
        if norm_first:
            raise NotImplementedError("norm_first=True (pre-LN) is not implemented; the reference default is post-LN")
        if embedding_width != hidden_size:
            raise NotImplementedError("embedding_width != hidden_size (factorised embeddings) is not implemented")
        if output_range is not None or embedding_layer is not None or with_dense_inputs:
            raise NotImplementedError("output_range / embedding_layer / with_dense_inputs are not implemented")
        self.name = name
        self._config = {
            "vocab_size": vocab_size, "hidden_size": hidden_size, "num_layers": num_layers,
            "num_attention_heads": num_attention_heads, "max_sequence_length": max_sequence_length,
            "inner_dim": inner_dim, "inner_activation": "gelu", "output_dropout": output_dropout,
            "attention_dropout": attention_dropout, "initializer": "TruncatedNormal(stddev=0.02)",
            "output_range": output_range, "embedding_width": embedding_width, "embedding_layer": embedding_layer,
            "norm_first": norm_first, "with_dense_inputs": with_dense_inputs,
        }
        cfg = make_model_config(vocab_size, hidden_size, num_layers, num_attention_heads, max_sequence_length, inner_dim,
                                output_dropout, attention_dropout)
        self.device = torch.device(device) if device is not None else _default_device()
        self.engine = Engine(cfg, self.device, seed=seed)   # raises ValueError on an unsupported geometry
        self.engine.init_parameters(seed=seed)
        self.inputs = {"input_word_ids": "int[B,L]", "input_mask": "int[B,L]"}

    # ---- Keras-like surface -----------------------------------------------------------------------------------------
    def __call__(self, inputs: Dict[str, torch.Tensor], training: bool = False) -> Dict[str, Any]:
        if not isinstance(inputs, dict):
            raise ValueError("Unexpected inputs type to %s." % self.__class__)
        enc_in = {"input_word_ids": inputs["input_word_ids"], "input_mask": inputs["input_mask"]}
        cb, keep = self.engine.prepare_batch(enc_in)
        self.engine.forward(cb, training=bool(training), pooler=True)
        return self._outputs(cb)

    call = __call__

    def _outputs(self, cb, copy: bool = True) -> Dict[str, Any]:
        """The output dict of Bert4RecEncoder.call (bert4rec_encoder.py:228-231).  The engine writes every forward of one batch
        shape into the same workspace; like the reference, a call returns tensors of its own (copy=True): results of two calls
        can be held and compared.  copy=False hands out views into the workspace (internal callers that consume them at once)."""
        B, L, P = cb.B, cb.L, cb.P
        H = self._config["hidden_size"]
        e = self.engine
        own = (lambda t: t.clone()) if copy else (lambda t: t)
        outs = [own(e.region(f"encoder_output_{i}", B, L, P).view(B, L, H)) for i in range(self._config["num_layers"])]
        return dict(sequence_output=outs[-1], pooled_output=own(e.region("pooled_output", B, L, P)), encoder_outputs=outs)

    def get_embedding_table(self) -> torch.Tensor:
        return self.engine.view("word_embeddings/embeddings")

    def get_config(self) -> dict:
        return dict(self._config)

    @classmethod
    def from_config(cls, config, custom_objects=None):
        return cls(**config)

    @property
    def trainable_variable_names(self):
        return [e.name for e in self.engine.table]
