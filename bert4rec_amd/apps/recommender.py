"""Single-sequence recommendation demo (mirrors bert4rec/apps/recommender.py:14-63): append a masked slot to the
history, forward, mask out already-seen items and the special tokens, return the argmax item."""
import numpy as np
import torch


class Recommender:
    def __init__(self, model, dataloader):
        self.model = model
        self.dataloader = dataloader

    def __call__(self, sequence: list, k: int = 1):
        tokenizer = self.dataloader.get_tokenizer()
        batch = self.dataloader.prepare_inference(list(sequence))
        batch = {key: torch.from_numpy(np.asarray(v)) for key, v in batch.items()}
        seen = set(tokenizer.tokenize(list(sequence))) | {0, 1, 2}
        cand = torch.tensor([[i for i in range(self.model.vocab_size) if i not in seen]], dtype=torch.int64)
        ranking, _, _, _ = self.model.rank_items_tensor(batch, cand)
        top = ranking[0, :k].cpu().tolist()
        items = tokenizer.detokenize(top)
        return items[0] if k == 1 else items
