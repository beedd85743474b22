from .components import networks  # noqa: F401
from .bert4rec_model import BERT4RecModel, History  # noqa: F401
from .model_wrapper import ModelWrapper  # noqa: F401
from .bert4rec_wrapper import BERT4RecModelWrapper  # noqa: F401
from . import model_utils  # noqa: F401
