"""Recommend the next item for a history (the reference's examples/recommender_app_example.py): load a saved model (run
bert4rec_ml_1m_example.py or bert4rec_lifecycle_example.py first), then ask the app."""
import pathlib
import sys

from _common import dataloaders, datasets, models

from bert4rec_amd.apps import Recommender
from bert4rec_amd.models import model_utils

if __name__ == "__main__":
    path = model_utils.determine_model_path(pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "bert4rec_ml-1m_lifecycle"))
    loaded = models.BERT4RecModelWrapper.load(path)
    kwargs = {"tokenizer": loaded["tokenizer"]} if "tokenizer" in loaded else {}
    if not datasets.ML1M.is_available():
        kwargs["data_source"] = datasets.synthetic_dataset(n_users=1500, n_items=3706, min_len=20, max_len=200, seed=0, order=0.6)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**kwargs)
    dataloader.generate_vocab()
    app = Recommender(loaded["model_wrapper"].model, dataloader)
    history = dataloader.get_tokenizer().detokenize([7, 19, 4, 33, 12])
    print("history:", history)
    print("next item:", app(history), " top 5:", app(history, k=5))
