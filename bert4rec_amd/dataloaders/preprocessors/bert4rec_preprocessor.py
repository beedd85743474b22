"""BERT4RecPreprocessor (mirrors bert4rec/dataloaders/preprocessors/bert4rec_preprocessor.py:11-168): truncation,
masking and padding of one user sequence into the six int64 arrays of the batch contract."""
import random

import numpy as np

from .base_preprocessor import BasePreprocessor
from .. import dataloader_utils


class BERT4RecPreprocessor(BasePreprocessor):
    tokenizer = None
    max_seq_len: int = None
    max_predictions_per_seq: int = None
    mask_token_id: int = None
    unk_token_id: int = None
    pad_token_id: int = None
    masked_lm_rate: float = None
    mask_token_rate: float = None
    random_token_rate: float = None

    @classmethod
    def set_properties(cls, tokenizer=None, max_seq_len=None, max_predictions_per_seq=None, mask_token_id=None,
                       unk_token_id=None, pad_token_id=None, masked_lm_rate=None, mask_token_rate=None,
                       random_token_rate=None):
        for k, v in dict(tokenizer=tokenizer, max_seq_len=max_seq_len, max_predictions_per_seq=max_predictions_per_seq,
                         mask_token_id=mask_token_id, unk_token_id=unk_token_id, pad_token_id=pad_token_id,
                         masked_lm_rate=masked_lm_rate, mask_token_rate=mask_token_rate,
                         random_token_rate=random_token_rate).items():
            if v is not None:
                setattr(cls, k, v)

    @classmethod
    def _window(cls, tokens: list, finetuning: bool) -> list:
        """bert4rec_preprocessor.py:61-67: the most recent max_seq_len tokens for finetuning / evaluation rows and for rows that
        fit; a random window of max_seq_len tokens of a longer training row (python `random`, as the reference)."""
        if finetuning or len(tokens) <= cls.max_seq_len:
            return tokens[-cls.max_seq_len:]
        start_i = random.randint(0, len(tokens) - cls.max_seq_len)
        return tokens[start_i:start_i + cls.max_seq_len]

    @classmethod
    def token_rows(cls, ds, finetuning: bool):
        """The dataset as a right-padded token matrix for on-device batch construction: per sequence tokenise + `_window`,
        no masking (that is b4r_mask_batch's part).  Returns a dataloader_utils.TokenMatrixDataset."""
        n = len(ds)
        rows = np.full((n, cls.max_seq_len), cls.pad_token_id, dtype=np.int64)
        for r, seq in enumerate(ds):
            seg = cls._window(cls.tokenizer.tokenize(seq), finetuning)
            rows[r, :len(seg)] = seg
        flags = np.full(n, 1 if finetuning else 0, dtype=np.int64)
        return dataloader_utils.TokenMatrixDataset(rows, flags, cls.max_predictions_per_seq, cls.tokenizer.get_vocab_size(),
                                                   cls.masked_lm_rate, cls.mask_token_rate, cls.random_token_rate)

    @classmethod
    def process_element(cls, sequence, apply_mlm: bool, finetuning: bool) -> dict:
        """bert4rec_preprocessor.py:48-116"""
        processed = dict()
        segments = cls._window(cls.tokenizer.tokenize(sequence), finetuning)
        input_word_ids = np.array(segments, dtype=np.int64)
        input_mask = np.ones_like(segments, dtype=np.int64)
        labels = input_word_ids.copy()
        if apply_mlm:
            if not finetuning:
                input_word_ids, pos, ids = dataloader_utils.apply_dynamic_masking_task(
                    input_word_ids, cls.max_predictions_per_seq, cls.mask_token_id, [cls.unk_token_id, cls.pad_token_id],
                    cls.tokenizer.get_vocab_size(), selection_rate=cls.masked_lm_rate,
                    mask_token_rate=cls.mask_token_rate, random_token_rate=cls.random_token_rate)
            else:
                input_word_ids, pos, ids = dataloader_utils.mask_last_token_only(input_word_ids, cls.mask_token_id)
            weights = np.ones_like(ids)
            padn = cls.max_predictions_per_seq - ids.shape[0]
            if padn > 0:
                ids, pos, weights = (np.pad(a, (0, padn), constant_values=cls.pad_token_id) for a in (ids, pos, weights))
            processed["masked_lm_ids"] = ids.astype(np.int64)
            processed["masked_lm_positions"] = pos.astype(np.int64)
            processed["masked_lm_weights"] = weights.astype(np.int64)
        padn = cls.max_seq_len - input_word_ids.shape[0]
        if padn > 0:
            input_word_ids, input_mask, labels = (np.pad(a, (0, padn), constant_values=cls.pad_token_id)
                                                  for a in (input_word_ids, input_mask, labels))
        processed["labels"] = labels
        processed["input_word_ids"] = input_word_ids
        processed["input_mask"] = input_mask
        return processed

    @classmethod
    def process_dataset(cls, ds, apply_mlm: bool = True, finetuning: bool = False):
        return dataloader_utils.ExampleDataset([cls.process_element(seq, apply_mlm, finetuning) for seq in ds])

    @classmethod
    def prepare_inference(cls, data) -> dict:
        """bert4rec_preprocessor.py:125-168: keep the most recent max_seq_len-1 items, append a placeholder that is
        masked as the position to predict; returns [1, .] arrays."""
        if type(data) is not list:
            raise ValueError("To prepare data for inference, please simply put in an unprocessed sequence of data "
                             "(i.e. a list of strings).")
        sequence = data[-cls.max_seq_len + 1:]
        sequence.append("[UNK]")
        out = cls.process_element(sequence, True, True)
        return {k: np.expand_dims(v, 0) for k, v in out.items()}
