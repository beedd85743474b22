import abc


class BasePreprocessor(abc.ABC):
    @classmethod
    @abc.abstractmethod
    def process_element(cls, sequence, **kwargs):
        pass
