"""mirrors bert4rec/dataloaders/samplers/base_sampler.py"""
import abc


class BaseSampler(abc.ABC):
    def __init__(self, source: list = None, vocab: list = None, sample_size: int = None):
        if sample_size is not None and sample_size < 0:
            raise ValueError(f"The sample size shouldn't be negative to avoid unexpected outputs (Given: {sample_size})")
        self.source = source.copy() if source is not None else None
        self.vocab = vocab.copy() if vocab is not None else None
        self.sample_size = sample_size

    def _get_parameters(self, source: list = None, vocab: list = None, sample_size: int = None):
        if source is None:
            source = self.source
        if vocab is None:
            vocab = self.vocab
        if sample_size is None:
            sample_size = self.sample_size
            if self.sample_size is None:
                raise ValueError("The sample size has to be given either during the initialization of the "
                                 "sampler or as an argument in the sample() method call.")
        if sample_size < 0:
            raise ValueError(f"A negative sample size is not allowed (Given: {sample_size})")
        return source, vocab, sample_size

    @abc.abstractmethod
    def sample(self, sample_size: int = None, source: list = None, vocab: list = None, without: list = None) -> list:
        pass

    @abc.abstractmethod
    def is_fully_prepared(self) -> bool:
        pass

    def set_source(self, source: list):
        self.source = source.copy()

    def set_vocab(self, vocab: list):
        self.vocab = vocab.copy()

    def set_sample_size(self, sample_size: int):
        self.sample_size = sample_size
