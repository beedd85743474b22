"""Rank a list of items for a history (the reference's examples/ranker_app_example.py).  Best item first (the reference's app
negates the logits before a descending sort and so returns the worst first: not reproduced)."""
import pathlib
import sys

from _common import dataloaders, datasets, models

from bert4rec_amd.apps import Ranker
from bert4rec_amd.models import model_utils

if __name__ == "__main__":
    path = model_utils.determine_model_path(pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "bert4rec_ml-1m_lifecycle"))
    loaded = models.BERT4RecModelWrapper.load(path)
    kwargs = {"tokenizer": loaded["tokenizer"]} if "tokenizer" in loaded else {}
    if not datasets.ML1M.is_available():
        kwargs["data_source"] = datasets.synthetic_dataset(n_users=1500, n_items=3706, min_len=20, max_len=200, seed=0, order=0.6)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**kwargs)
    dataloader.generate_vocab()
    tok = dataloader.get_tokenizer()
    app = Ranker(loaded["model_wrapper"].model, dataloader)
    history, items = tok.detokenize([7, 19, 4, 33, 12]), tok.detokenize([40, 41, 42, 43, 44, 45])
    print("history:", history, "\ncandidates:", items, "\nranked:", app(history, items))
