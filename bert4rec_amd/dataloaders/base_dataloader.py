import abc
from typing import Type, Union

from .. import tokenizers


class BaseDataloader(abc.ABC):
    def __init__(self, tokenizer: Union[str, "tokenizers.BaseTokenizer"] = None, data_source: Type = None,
                 preprocessor: Type = None, **kwargs):
        self.tokenizer = tokenizers.get(tokenizer)
        self.data_source = data_source
        self.preprocessor = preprocessor

    def get_tokenizer(self):
        return self.tokenizer

    @property
    @abc.abstractmethod
    def dataset_identifier(self):
        pass
