"""Single-sequence recommendation (the reference's demo app, bert4rec/apps/recommender.py:14-63): append a masked slot to the
history (prepare_inference), score the whole vocabulary for that slot and return the best item(s) the user has not seen.

On the GPU this is the evaluation path: encoder forward, tfm MaskedLM's transform on the ONE masked slot, one
b4r_rank_candidates call over the whole vocabulary (cand = NULL: no [1, V] candidate list, no [B, P, V] logits).  Deliberate
differences from the reference, both documented in INTEGRATION.md: it reads ``mlm_logits[:, -1]``, i.e. the LAST of the P slots
-- a padded slot that gathers position 0 -- where the masked token sits in slot 0 (used here); and it can return [PAD] / [MASK] /
[UNK], which are excluded here together with the seen items."""
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
        blocked = set(tokenizer.tokenize(list(sequence))) | {0, 1, 2}
        ranking, _, _, _ = self.model.rank_items_tensor(batch, None)      # [1, V]: the whole vocabulary, best first
        # the best k ids outside `blocked` are among the first k + |blocked| entries
        head = ranking[0, :k + len(blocked)].cpu().tolist()
        top = [i for i in head if i not in blocked][:k]
        items = tokenizer.detokenize(top)
        return items[0] if k == 1 else items
