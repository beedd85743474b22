from .recommender import Recommender  # noqa: F401
from .ranker import Ranker  # noqa: F401
