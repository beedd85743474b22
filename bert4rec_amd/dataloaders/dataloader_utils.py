"""Batch construction utilities (mirror bert4rec/dataloaders/dataloader_utils.py).  Integer work on the host; the
outputs are the int64 batch dict the HIP path consumes.

Containers replace tf.data.Dataset: `SequenceDataset` (ragged python sequences per user), `ExampleDataset`
(per-example dict of fixed-length int64 arrays) and `BatchedDataset` (list of batch dicts of torch int64 tensors; like
the reference's `.cache()` after shuffle+batch, the batches -- masks and order -- are fixed once built)."""
from __future__ import annotations

import collections
import random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd
import torch


# ---------------------------------------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------------------------------------
class SequenceDataset:
    """One python list (the user's item sequence) per element."""

    def __init__(self, sequences: Sequence[Sequence]):
        self.sequences = [list(s) for s in sequences]

    def __len__(self):
        return len(self.sequences)

    def __iter__(self):
        return iter(self.sequences)

    def cardinality(self) -> int:
        return len(self.sequences)

    def repeat(self, n: int) -> "SequenceDataset":
        return SequenceDataset(self.sequences * n)

    def take(self, n: int) -> "SequenceDataset":
        return SequenceDataset(self.sequences[:n])

    def skip(self, n: int) -> "SequenceDataset":
        return SequenceDataset(self.sequences[n:])

    def shuffle(self, seed=None) -> "SequenceDataset":
        idx = np.random.RandomState(seed).permutation(len(self.sequences))
        return SequenceDataset([self.sequences[i] for i in idx])

    def concatenate(self, other: "SequenceDataset") -> "SequenceDataset":
        return SequenceDataset(self.sequences + other.sequences)


class ExampleDataset:
    """Processed examples: a list of dicts of 1-D int64 numpy arrays (the 6 keys of the batch contract)."""

    def __init__(self, examples: List[Dict[str, np.ndarray]]):
        self.examples = examples

    def __len__(self):
        return len(self.examples)

    def __iter__(self):
        return iter(self.examples)

    def cardinality(self) -> int:
        return len(self.examples)

    def concatenate(self, other: "ExampleDataset") -> "ExampleDataset":
        return ExampleDataset(self.examples + other.examples)

    def take(self, n: int) -> "ExampleDataset":
        return ExampleDataset(self.examples[:n])

    def skip(self, n: int) -> "ExampleDataset":
        return ExampleDataset(self.examples[n:])

    def shuffle(self, seed=None) -> "ExampleDataset":
        idx = np.random.RandomState(seed).permutation(len(self.examples))
        return ExampleDataset([self.examples[i] for i in idx])


class ResidentBatch(dict):
    """A batch dict whose tensors stay in HBM between epochs (the six tensors of the reference's batches, nothing else, as keys) plus
    per-batch constants as ATTRIBUTES: slot_index [R, 2] = the (row, slot) pairs with masked_lm_weights != 0, found once on the host
    copy or when the batch is frozen -- the evaluator otherwise looks for them on the device with a read-back per batch, which makes
    every batch wait for the kernels of the one before; eval_cache = what the evaluator derives from them on its first pass."""
    __slots__ = ("slot_index", "eval_cache")

    def __init__(self, tensors, slot_index=None):
        super().__init__(tensors)
        self.slot_index = slot_index
        self.eval_cache = None


class BatchedDataset:
    """List of batch dicts (torch int64 [B, .]).  Iterating yields the same batches every epoch (== tf .cache())."""

    def __init__(self, batches: List[Dict[str, torch.Tensor]]):
        self.batches = batches
        self._device_batches = None
        self._device = None

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self._device_batches if self._device_batches is not None else self.batches)

    def cardinality(self) -> int:
        return len(self.batches)

    def cache_on_device(self, device) -> "BatchedDataset":
        """Keep a copy of every batch in HBM (a whole ML-1M epoch is ~200 MB): no per-step host->device traffic."""
        device = torch.device(device)
        if device.type != "cuda":
            return self
        if self._device_batches is None or self._device != device:
            self._device_batches = []
            for hb in self.batches:
                idx = None
                if "masked_lm_weights" in hb:   # found on the host copy: no device read-back
                    w = torch.as_tensor(hb["masked_lm_weights"])
                    idx = torch.nonzero(w.reshape(w.shape[0], -1) != 0).to(device, non_blocking=True)
                self._device_batches.append(ResidentBatch({k: v.to(device, non_blocking=True) for k, v in hb.items()}, idx))
            self._device = device
        return self

    def host_batches(self):
        return self.batches


class TokenMatrixDataset:
    """A dataset as ONE right-padded int64 token matrix [U, L] plus a per-row "last-token mask" flag: what on-device batch
    construction starts from.  Rows are already truncated by the preprocessor's rule
    (bert4rec_preprocessor.py:61-67); masking happens per batch on the GPU (b4r_mask_batch)."""

    def __init__(self, tokens: np.ndarray, finetune_rows: np.ndarray, max_predictions_per_seq: int, vocab_size: int,
                 selection_rate: float, mask_token_rate: float, random_token_rate: float):
        self.tokens = np.ascontiguousarray(tokens, dtype=np.int64)
        self.finetune_rows = np.ascontiguousarray(finetune_rows, dtype=np.int64)
        if self.tokens.ndim != 2 or self.finetune_rows.shape != (self.tokens.shape[0],):
            raise ValueError("tokens must be [rows, length] with one finetune flag per row")
        self.max_predictions_per_seq = int(max_predictions_per_seq)
        self.vocab_size = int(vocab_size)
        self.selection_rate, self.mask_token_rate, self.random_token_rate = selection_rate, mask_token_rate, random_token_rate

    def __len__(self):
        return self.tokens.shape[0]

    def cardinality(self) -> int:
        return len(self)

    def _like(self, tokens, flags) -> "TokenMatrixDataset":
        return TokenMatrixDataset(tokens, flags, self.max_predictions_per_seq, self.vocab_size, self.selection_rate,
                                  self.mask_token_rate, self.random_token_rate)

    def concatenate(self, other: "TokenMatrixDataset") -> "TokenMatrixDataset":
        return self._like(np.concatenate([self.tokens, other.tokens]), np.concatenate([self.finetune_rows, other.finetune_rows]))

    def take(self, n: int) -> "TokenMatrixDataset":
        return self._like(self.tokens[:n], self.finetune_rows[:n])

    def skip(self, n: int) -> "TokenMatrixDataset":
        return self._like(self.tokens[n:], self.finetune_rows[n:])


def bucket_order(order: np.ndarray, lengths: np.ndarray, batch_size: int, bucket_batches: int, seed: int) -> np.ndarray:
    """Length bucketing: every window of `bucket_batches` consecutive batches of the (already shuffled) order is sorted by sequence
    length (stable), so that a batch holds sequences of similar length and `trim_padding` can cut the padding they share; the
    batches of a window are then put back in a random order.  bucket_batches <= 1: the order as it is."""
    order = np.asarray(order, dtype=np.int64)
    if bucket_batches <= 1:
        return order
    rng = np.random.RandomState(seed)
    out = []
    win = batch_size * bucket_batches
    for s in range(0, len(order), win):
        chunk = order[s:s + win]
        chunk = chunk[np.argsort(lengths[chunk], kind="stable")]
        parts = [chunk[b:b + batch_size] for b in range(0, len(chunk), batch_size)]
        full = [p for p in parts if len(p) == batch_size]
        rest = [p for p in parts if len(p) != batch_size]      # the data set's partial batch stays last
        out.extend(full[i] for i in rng.permutation(len(full)))
        out.extend(rest)
    return np.concatenate(out) if out else order


def trimmed_length(longest: int, full: int, multiple: int = 16) -> int:
    """Columns a batch keeps under trim_padding: its longest sequence rounded up to a multiple of 16 (one token tile)."""
    return int(min(full, max(multiple, -(-int(longest) // multiple) * multiple)))


class DeviceMaskedBatches:
    """Batches built on the GPU from a TokenMatrixDataset: the token matrix lives in HBM, a batch is an index list and one
    b4r_mask_batch launch.  `remask_each_epoch=False` (default) reproduces the reference, whose `.cache()` behind shuffle + batch
    freezes batch composition AND masks after the first epoch (dataloader_utils.py:341-346): the batches of the first pass are
    kept.  True draws new masks every epoch (same batch composition) -- what the duplication factor approximates on the host.
    `trim_padding`: a batch keeps only the columns its longest sequence needs (make_batches)."""

    def __init__(self, dataset: TokenMatrixDataset, order: np.ndarray, batch_size: int, seed: int, remask_each_epoch: bool,
                 trim_padding: bool = False):
        self.dataset, self.order, self.batch_size = dataset, np.asarray(order, dtype=np.int64), int(batch_size)
        self.seed, self.remask_each_epoch = int(seed), bool(remask_each_epoch)
        self.trim_padding = bool(trim_padding)
        self.epoch = 0
        self._device = None
        self._tokens = self._flags = self._order = None
        self._frozen = None
        # columns per batch, known on the host (no device round trip while an epoch runs)
        full = dataset.tokens.shape[1]
        lens = sequence_lengths(dataset.tokens)
        self.batch_columns = [trimmed_length(lens[self.order[s:s + self.batch_size]].max(), full) if self.trim_padding else full
                              for s in range(0, len(self.order), self.batch_size)]
        # masked-LM slots a row can use: the count rule of apply_dynamic_masking_task on its length (an upper bound: special tokens
        # inside a row only lower the count), one slot for a last-token row
        P = dataset.max_predictions_per_seq
        slots = np.where(dataset.finetune_rows != 0, 1, np.minimum(P, np.maximum(1, (lens * dataset.selection_rate).astype(np.int64))))
        self.batch_slots = [trimmed_length(slots[self.order[s:s + self.batch_size]].max(), P, 4) if self.trim_padding else P
                            for s in range(0, len(self.order), self.batch_size)]

    def __len__(self):
        return (len(self.order) + self.batch_size - 1) // self.batch_size

    def cardinality(self) -> int:
        return len(self)

    def cache_on_device(self, device) -> "DeviceMaskedBatches":
        device = torch.device(device)
        if self._device != device:
            self._device = device
            self._tokens = torch.from_numpy(self.dataset.tokens).to(device)
            self._flags = torch.from_numpy(self.dataset.finetune_rows).to(device)
            self._order = torch.from_numpy(self.order).to(device)
            self._frozen = None
        return self

    def _build(self, epoch: int):
        from ..engine import device_mask_batch
        if self._device is None:
            self.cache_on_device("cuda")
        ds = self.dataset
        full = ds.tokens.shape[1]
        for j, s in enumerate(range(0, len(self.order), self.batch_size)):
            batch = device_mask_batch(self._tokens, ds.max_predictions_per_seq, ds.vocab_size, ds.selection_rate, ds.mask_token_rate,
                                      ds.random_token_rate, False, (self.seed << 20) + epoch, self._order[s:s + self.batch_size],
                                      self._flags, self._device)
            cols, slots = self.batch_columns[j], self.batch_slots[j]
            if cols < full or slots < ds.max_predictions_per_seq:
                batch = {k: v[:, :(cols if k in PER_TOKEN_KEYS else slots)].contiguous() for k, v in batch.items()}
            yield batch

    def __iter__(self):
        if self.remask_each_epoch:
            epoch, self.epoch = self.epoch, self.epoch + 1
            return self._build(epoch)
        if self._frozen is None:
            # batches that stay: the (row, slot) pairs with a weight are found once, here (one read-back per batch, at build time)
            self._frozen = [ResidentBatch(b, torch.nonzero(b["masked_lm_weights"] != 0)) for b in self._build(0)]
        return iter(self._frozen)


PER_TOKEN_KEYS = ("input_word_ids", "input_mask", "labels")


def sequence_lengths(tokens: np.ndarray) -> np.ndarray:
    """Length of every right-padded row: one past its last token that is not [PAD] (id 0)."""
    nz = tokens != 0
    return np.where(nz.any(1), tokens.shape[1] - np.argmax(nz[:, ::-1], axis=1), 0).astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# reference utilities
# ---------------------------------------------------------------------------------------------------------------------
def rank_items_by_popularity(items: list) -> list:
    """dataloader_utils.py:14-18: most frequent first, ties keep first-occurrence order, duplicates removed."""
    sorted_item_list = sorted(items, key=collections.Counter(items).get, reverse=True)
    return list(dict.fromkeys(sorted_item_list))


def duplicate_dataset(ds: SequenceDataset, duplication_factor: int) -> SequenceDataset:
    """dataloader_utils.py:177-183"""
    if duplication_factor < 1:
        raise ValueError(f"A duplication factor of less than 1 (given: {duplication_factor}) is not allowed!")
    return ds.repeat(duplication_factor) if duplication_factor > 1 else ds


def make_sequence_df(df: pd.DataFrame, group_column_name: str, extract_sequences: list,
                     min_sequence_length: int = 0) -> pd.DataFrame:
    rows = []
    for _, g in df.groupby(group_column_name):
        row = {}
        ok = True
        for col in extract_sequences:
            seq = g[col].to_list()
            if len(seq) < min_sequence_length:
                ok = False
                break
            row[col] = seq
        if ok:
            rows.append(row)
    return pd.DataFrame(rows)


def split_sequence_df(df: pd.DataFrame, group_by_column: str, extract_columns: list,
                      min_sequence_length: int = 5) -> Tuple[pd.DataFrame, pd.DataFrame, pd.DataFrame]:
    """dataloader_utils.py:113-174: per user, train = first n-2, val = first n-1, test = all n items, only when
    n >= min_sequence_length; shorter sequences go to train only (whole)."""
    if group_by_column not in df.columns:
        raise ValueError(f"Group column key {group_by_column} is not present in columns in dataframe: {df.columns}")
    if len(extract_columns) - 1 > len(df.columns):
        raise ValueError("More columns to extract have been given than there are actual columns in the dataframe: "
                         f"{len(df.columns)}")
    for col in extract_columns:
        if col not in df.columns:
            raise ValueError(f"Column key {col} of the extract_columns argument is not present in columns in "
                             f"dataframe: {df.columns}")
    train, val, test = {}, {}, {}
    for i, (_, g) in enumerate(df.groupby(group_by_column)):
        train[i], val[i], test[i] = {}, {}, {}
        for col in extract_columns:
            seq = g[col].to_list()
            train[i][col] = seq
            if len(seq) >= min_sequence_length:
                train[i][col] = seq[:-2]
                val[i][col] = seq[:-1]
                test[i][col] = seq
    to_df = lambda d: pd.DataFrame.from_dict(d, orient="index")
    return to_df(train), to_df(val), to_df(test)


def convert_df_to_ds(df: pd.DataFrame, datatypes: list = None) -> SequenceDataset:
    """Single sequence column -> SequenceDataset (rows that are NaN -- users too short for val/test -- are dropped)."""
    if datatypes is not None and len(datatypes) != len(df.columns):
        raise ValueError(f"The given datatypes list ({datatypes}, len: {len(datatypes)}) has to have as many elements "
                         f"as columns in the given df ({len(df.columns)}).")
    if len(df.columns) == 0:
        return SequenceDataset([])
    col = df[df.columns[0]]
    return SequenceDataset([s for s in col.to_list() if isinstance(s, (list, tuple, np.ndarray))])


def apply_dynamic_masking_task(sequence: np.ndarray, max_selections_per_seq: int, mask_token_id: int,
                               special_token_ids: List[int], vocab_size: int, selection_rate: float = 0.2,
                               mask_token_rate: float = 0.8, random_token_rate: float = 0.1,
                               seed: int = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """dataloader_utils.py:186-261.  Uses python's `random` in the same call order as the reference, so equal seeds give
    equal outputs (pinned by tests/golden/reference_goldens.json)."""
    dtype = sequence.dtype
    random.seed(seed)
    n_plain = int((~np.isin(sequence, special_token_ids)).sum())
    num_to_predict = min(max_selections_per_seq, max(1, int(n_plain * selection_rate)))
    selectable_vocab = None
    pos_indexes = list(range(n_plain))
    random.shuffle(pos_indexes)
    pos_indexes = sorted(pos_indexes[:num_to_predict])
    masked_token_ids = sequence.copy()
    ids, positions = [], []
    for index in pos_indexes:
        if len(ids) >= num_to_predict:
            break
        replaced_token = sequence[index]
        rn = random.random()
        if rn < mask_token_rate + random_token_rate:
            if selectable_vocab is None:
                selectable_vocab = [i for i in range(vocab_size) if i not in special_token_ids]
            replaced_token = random.choice(selectable_vocab)
        if rn < mask_token_rate:
            replaced_token = mask_token_id
        masked_token_ids[index] = replaced_token
        ids.append(sequence[index])
        positions.append(index)
    return masked_token_ids, np.array(positions, dtype=dtype), np.array(ids, dtype=dtype)


def mask_last_token_only(sequence: np.ndarray, mask_token_id: int):
    """dataloader_utils.py:264-269"""
    seq = np.array(sequence, dtype=np.int64)
    masked_lm_ids = np.array([seq[-1]], dtype=np.int64)
    seq[-1] = mask_token_id
    return seq, np.array([len(seq) - 1], dtype=np.int64), masked_lm_ids


def split_dataset(ds, ds_size: int = None, train_split: float = 0.8, val_split: float = 0.1, test_split: float = 0.1,
                  shuffle: bool = True, shuffle_size: int = 10000, seed: int = 12):
    """dataloader_utils.py:272-303 (the shuffle is a seeded permutation; tf's stream cannot be reproduced)."""
    if (train_split + test_split + val_split) != 1:
        raise ValueError("The dataset can only be split in parts that sum up to 1 or a 100%.")
    if ds_size is None:
        ds_size = len(ds)
    if shuffle:
        ds = ds.shuffle(seed=seed)
    train_size = int(train_split * ds_size)
    val_size = int(val_split * ds_size)
    return ds.take(train_size), ds.skip(train_size).take(val_size), ds.skip(train_size).skip(val_size)


def make_batches(dataset, buffer_size: int = None, batch_size: int = 64, squeeze_tensors: bool = False,
                 reshuffle_each_iteration: bool = False, seed: int = None, remask_each_epoch: bool = False,
                 bucket_by_length: int = 0, trim_padding: bool = False):
    """dataloader_utils.py:306-346: shuffle(all) -> batch (last batch may be partial) -> cache.  Because the reference
    caches AFTER shuffle+batch, batch composition and masks are frozen after the first epoch; so are they here.
    A TokenMatrixDataset (prepare_training(device_masking=True)) gives batches that are masked on the GPU; only there
    `remask_each_epoch=True` is available (new masks per epoch, same composition).
    Not in the reference (it pads every row to max_seq_len, bert4rec_preprocessor.py:105-110), both off by default:
    `trim_padding=True` cuts every batch to the columns its longest sequence needs (a multiple of 16) and to the masked-LM slots
    its rows can use (a multiple of 4) -- padded keys are masked, padded positions and padded slots carry no loss, so loss,
    masked accuracy and gradients are unchanged while the kernels skip the common padding (only `sparse_categorical_accuracy`,
    which the reference averages over ALL slots including the padded ones, now averages over the slots that are kept);
    `bucket_by_length=W` (> 1) sorts every window of W batches by length first, so that the batches have padding to cut."""
    if reshuffle_each_iteration:
        raise NotImplementedError("reshuffle_each_iteration has no effect behind the reference's .cache(); not offered")
    n = len(dataset)
    order = np.random.RandomState(seed).permutation(n)
    if isinstance(dataset, TokenMatrixDataset):
        order = bucket_order(order, sequence_lengths(dataset.tokens), batch_size, bucket_by_length, 0 if seed is None else seed)
        return DeviceMaskedBatches(dataset, order, batch_size, 0 if seed is None else seed, remask_each_epoch, trim_padding)
    if remask_each_epoch:
        raise ValueError("remask_each_epoch needs a dataset prepared with device_masking=True")
    lens = np.array([int(np.asarray(e["input_mask"]).sum()) for e in dataset.examples], dtype=np.int64)
    order = bucket_order(order, lens, batch_size, bucket_by_length, 0 if seed is None else seed)
    batches = []
    for s in range(0, n, batch_size):
        idx = order[s:s + batch_size]
        keys = dataset.examples[idx[0]].keys()
        batch = {k: torch.from_numpy(np.stack([dataset.examples[i][k] for i in idx]).astype(np.int64)) for k in keys}
        if trim_padding:
            full = batch["input_word_ids"].shape[1]
            cols = trimmed_length(lens[idx].max(), full)
            slots = batch["masked_lm_weights"].shape[1] if "masked_lm_weights" in batch else 0
            if slots:
                slots = trimmed_length(int((batch["masked_lm_weights"] != 0).sum(1).max()), slots, 4)
            batch = {k: v[:, :(cols if k in PER_TOKEN_KEYS else slots)].contiguous() for k, v in batch.items()}
        batches.append(batch)
    return BatchedDataset(batches)
