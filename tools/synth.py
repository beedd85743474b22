"""Synthetic batches for the measurement helpers (S-full rows, S-ragged rows with lengths ~ U{5..L}, S-eval rows
with only the last item masked).  Deterministic per seed; the same row rules as bench.py's generator."""
import numpy as np
import torch

MASK_ID = 1


def synthetic_batch(B, L, P, V, seed=0, rate=0.2, ragged=False, finetune=False):
    rng = np.random.default_rng(seed)
    out = {k: np.zeros((B, L), np.int64) for k in ("input_word_ids", "input_mask", "labels")}
    out.update({k: np.zeros((B, P), np.int64) for k in ("masked_lm_positions", "masked_lm_ids", "masked_lm_weights")})
    for b in range(B):
        n = int(rng.integers(5, L + 1)) if ragged else L
        items = rng.integers(3, V, size=n)
        where = np.array([n - 1]) if finetune else np.sort(rng.choice(n, size=min(P, max(1, int(n * rate))), replace=False))
        out["labels"][b, :n] = items
        out["input_mask"][b, :n] = 1
        out["masked_lm_positions"][b, :len(where)] = where
        out["masked_lm_ids"][b, :len(where)] = items[where]
        out["masked_lm_weights"][b, :len(where)] = 1
        items = items.copy()
        items[where] = MASK_ID
        out["input_word_ids"][b, :n] = items
    return {k: torch.from_numpy(v) for k, v in out.items()}
