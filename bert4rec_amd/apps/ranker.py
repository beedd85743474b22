"""Rank a given list of items for one user history (mirrors bert4rec/apps/ranker.py:19-76).  The reference negates the
logits before a DESCENDING sort (ranker.py:29), i.e. it returns the worst item first: a bug that is not reproduced
here -- the best item comes first."""
import numpy as np
import torch


class Ranker:
    def __init__(self, model, dataloader):
        self.model = model
        self.dataloader = dataloader

    def __call__(self, sequence: list, items: list):
        tokenizer = self.dataloader.get_tokenizer()
        batch = self.dataloader.prepare_inference(list(sequence))
        batch = {key: torch.from_numpy(np.asarray(v)) for key, v in batch.items()}
        cand = torch.tensor([tokenizer.tokenize(list(items))], dtype=torch.int64)
        ranking, _, _, _ = self.model.rank_items_tensor(batch, cand)
        return tokenizer.detokenize(ranking[0].cpu().tolist())
