"""Tokenizer factory (mirrors bert4rec/tokenizers/__init__.py)."""
from typing import Union

from .base_tokenizer import BaseTokenizer
from .simple_tokenizer import SimpleTokenizer

tokenizers_map = {"simple": SimpleTokenizer}


def get(identifier: Union[str, BaseTokenizer] = "simple", **kwargs) -> BaseTokenizer:
    if isinstance(identifier, str) and identifier in tokenizers_map:
        return tokenizers_map[identifier](**kwargs)
    if isinstance(identifier, BaseTokenizer):
        return identifier
    raise ValueError(f"{identifier} is not known!")
