"""Tokenizers by name.  `get("simple", **kwargs)` builds one, `get(instance)` passes it through — the call shapes of
bert4rec/tokenizers/__init__.py; `tokenizers_map` is the public registry a caller may extend."""
from typing import Union

from .base_tokenizer import BaseTokenizer
from .simple_tokenizer import SimpleTokenizer

tokenizers_map = {"simple": SimpleTokenizer}


def get(identifier: Union[str, BaseTokenizer] = "simple", **kwargs) -> BaseTokenizer:
    if isinstance(identifier, BaseTokenizer):
        return identifier                    # already built: keyword arguments do not apply
    cls = tokenizers_map.get(identifier) if isinstance(identifier, str) else None
    if cls is None:
        raise ValueError(f"{identifier} is not known!")
    return cls(**kwargs)
