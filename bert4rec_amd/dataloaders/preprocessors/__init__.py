from .base_preprocessor import BasePreprocessor  # noqa: F401
from .bert4rec_preprocessor import BERT4RecPreprocessor  # noqa: F401
