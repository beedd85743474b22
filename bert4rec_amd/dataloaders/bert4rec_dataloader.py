"""BERT4RecDataloader and its five dataset flavours (mirror bert4rec/dataloaders/bert4rec_dataloader.py:12-230 and
bert4rec_{ml1m,ml20m,steam,beauty,reddit}_dataloader.py, which differ only in defaults and column names)."""
from typing import Type, Union

from .. import datasets, tokenizers
from . import dataloader_utils as utils
from . import preprocessors
from .base_dataloader import BaseDataloader


class BERT4RecDataloader(BaseDataloader):
    # column conventions of the concrete dataset (overridden below)
    _SORT_BY = None
    _ITEM_COLUMN = None
    _GROUP_BY = None
    _IDENTIFIER = None

    def __init__(self, max_seq_len: int, max_predictions_per_seq: int,
                 tokenizer: Union[str, tokenizers.BaseTokenizer] = "simple", data_source: Type = None,
                 preprocessor: Type = preprocessors.BERT4RecPreprocessor, masked_lm_prob: float = 0.2,
                 mask_token_rate: float = 1.0, random_token_rate: float = 0.0, input_duplication_factor: int = 1,
                 min_sequence_len: int = 5):
        tokenizer = tokenizers.get(tokenizer)
        super().__init__(tokenizer, data_source, preprocessor)
        if input_duplication_factor < 1:
            raise ValueError("An input_duplication_factor of less than 1 is not allowed!")
        self._PAD_TOKEN, self._MASK_TOKEN, self._UNK_TOKEN = "[PAD]", "[MASK]", "[UNK]"
        # insertion order fixes the ids: PAD = 0, MASK = 1, UNK = 2 (bert4rec_dataloader.py:38-43)
        self._PAD_TOKEN_ID = self.tokenizer.tokenize(self._PAD_TOKEN)
        self._MASK_TOKEN_ID = self.tokenizer.tokenize(self._MASK_TOKEN)
        self._UNK_TOKEN_ID = self.tokenizer.tokenize(self._UNK_TOKEN)
        self._SPECIAL_TOKENS = [self._PAD_TOKEN, self._UNK_TOKEN, self._MASK_TOKEN]
        self._SPECIAL_TOKEN_IDS = [self._PAD_TOKEN_ID, self._MASK_TOKEN_ID, self._UNK_TOKEN_ID]
        self._MAX_PREDICTIONS_PER_SEQ = max_predictions_per_seq
        self._MAX_SEQ_LENGTH = max_seq_len
        self.masked_lm_prob = masked_lm_prob
        self.mask_token_rate = mask_token_rate
        self.random_token_rate = random_token_rate
        self.input_duplication_factor = input_duplication_factor
        self.min_sequence_len = min_sequence_len
        self._df = None

    @property
    def dataset_identifier(self):
        if self._IDENTIFIER is None:
            raise NotImplementedError("The dataset_identifier method hasn't been implemented.")
        return self._IDENTIFIER

    def _frame(self):
        if self._df is None:
            if self.data_source is None:
                raise ValueError("this dataloader has no data_source")
            self._df = self.data_source.load_data()
        return self._df

    def _set_preprocessor_properties(self):
        self.preprocessor.set_properties(tokenizer=self.tokenizer, max_seq_len=self._MAX_SEQ_LENGTH,
                                         max_predictions_per_seq=self._MAX_PREDICTIONS_PER_SEQ,
                                         mask_token_id=self._MASK_TOKEN_ID, unk_token_id=self._UNK_TOKEN_ID,
                                         pad_token_id=self._PAD_TOKEN_ID, masked_lm_rate=self.masked_lm_prob,
                                         mask_token_rate=self.mask_token_rate, random_token_rate=self.random_token_rate)

    def load_data(self, split_data: bool = True, sort_by: str = None, extract_data: list = None, datatypes: list = None,
                  duplication_factor: int = None, group_by: str = None) -> tuple:
        """bert4rec_dataloader.py:115-142"""
        sort_by = sort_by if sort_by is not None else self._SORT_BY
        extract_data = extract_data if extract_data else [self._ITEM_COLUMN]
        datatypes = datatypes if datatypes else ["list"]
        group_by = group_by if group_by is not None else self._GROUP_BY
        if len(extract_data) != len(datatypes):
            raise ValueError(f"The length of the extract_data list ({len(extract_data)}) has to be the same as the length "
                             f"of the datatypes list ({len(datatypes)}).")
        df = self._frame()
        if sort_by is not None:
            df = df.sort_values(by=sort_by)
        if not split_data:
            dfs = (utils.make_sequence_df(df, group_column_name=group_by, extract_sequences=extract_data),)
        else:
            dfs = utils.split_sequence_df(df, group_by, extract_data, self.min_sequence_len)
        dss = [utils.convert_df_to_ds(d, datatypes) for d in dfs]
        if duplication_factor is None:
            duplication_factor = self.input_duplication_factor
        dss[0] = utils.duplicate_dataset(dss[0], duplication_factor)
        return tuple(dss)

    def process_data(self, ds, apply_mlm: bool = True, finetuning: bool = False, device_masking: bool = False):
        """device_masking: only tokenise + truncate on the host (one [U, L] matrix); the masked-LM task and the padding of the
        six tensors run per batch on the GPU (b4r_mask_batch via dataloader_utils.make_batches)."""
        self._set_preprocessor_properties()
        if device_masking:
            if not apply_mlm:
                raise ValueError("device_masking builds masked-LM batches; use apply_mlm=True")
            return self.preprocessor.token_rows(ds, finetuning)
        return self.preprocessor.process_dataset(ds, apply_mlm, finetuning)

    def get_data(self, split_data: bool = True, sort_by: str = None, extract_data: list = None, datatypes: list = None,
                 duplication_factor: int = None, group_by: str = None, apply_mlm: bool = True,
                 finetuning_split: float = 0, device_masking: bool = False) -> tuple:
        """bert4rec_dataloader.py:56-113: validation/test (datasets 1,2) always use the last-token mask; a
        `finetuning_split` share of the training examples does too (split_dataset seed 12)."""
        if finetuning_split < 0 or finetuning_split > 1:
            raise ValueError(f"The finetuning_split argument has to be a float between 0 and 1. Given: {finetuning_split}")
        dss = self.load_data(split_data, sort_by, extract_data, datatypes, duplication_factor, group_by)
        processed = []
        for i, ds in enumerate(dss):
            dm = device_masking
            if i >= 1:
                processed.append(self.process_data(ds, apply_mlm, finetuning=True, device_masking=dm))
            elif finetuning_split > 0:
                train_ds, ft_ds, _ = utils.split_dataset(ds, train_split=1 - finetuning_split,
                                                         val_split=finetuning_split, test_split=0.0)
                processed.append(self.process_data(train_ds, finetuning=False, device_masking=dm)
                                 .concatenate(self.process_data(ft_ds, finetuning=True, device_masking=dm)))
            else:
                processed.append(self.process_data(ds, apply_mlm, finetuning=False, device_masking=dm))
        return tuple(processed)

    def generate_vocab(self, source=None, progress_bar: bool = True) -> bool:
        if source is None:
            # the reference tokenizes set(df[item]) (arbitrary set order); a sorted set makes ids reproducible
            source = sorted(set(self._frame()[self._ITEM_COLUMN]))
        self.tokenizer.tokenize(source, progress_bar)
        return True

    def prepare_training(self, sort_by: str = None, extract_data: list = None, datatypes: list = None,
                         group_by: str = None, finetuning_split: float = 0.1, device_masking: bool = False) -> tuple:
        """bert4rec_dataloader.py:144-158.  device_masking=True: the three datasets come back as token matrices
        (dataloader_utils.TokenMatrixDataset) whose batches are masked on the GPU by make_batches; with
        make_batches(..., remask_each_epoch=True) every epoch sees new masks, so the host-side `input_duplication_factor`
        (ten masked copies of ML-1M) can be 1."""
        if finetuning_split < 0 or finetuning_split > 1:
            raise ValueError(f"The finetuning_split argument has to be a float between 0 and 1. Given: {finetuning_split}")
        self.generate_vocab()
        return self.get_data(split_data=True, sort_by=sort_by, extract_data=extract_data, datatypes=datatypes,
                             group_by=group_by, finetuning_split=finetuning_split, apply_mlm=True,
                             device_masking=device_masking)

    def prepare_inference(self, data):
        self._set_preprocessor_properties()
        return self.preprocessor.prepare_inference(data)

    def create_item_list(self) -> list:
        return self._frame()[self._ITEM_COLUMN].to_list()

    def create_item_list_tokenized(self) -> list:
        return self.tokenizer.tokenize(self.create_item_list())


def _flavour(identifier, data_source, sort_by, item_col, group_by, L, P, prob, dup, min_len):
    class _Loader(BERT4RecDataloader):
        _SORT_BY, _ITEM_COLUMN, _GROUP_BY, _IDENTIFIER = sort_by, item_col, group_by, identifier

        def __init__(self, max_seq_len: int = L, max_predictions_per_seq: int = P, tokenizer="simple",
                     data_source: Type = data_source, preprocessor: Type = preprocessors.BERT4RecPreprocessor,
                     masked_lm_prob: float = prob, mask_token_rate: float = 1.0, random_token_rate: float = 0.0,
                     input_duplication_factor: int = dup, min_sequence_len: int = min_len):
            super().__init__(max_seq_len, max_predictions_per_seq, tokenizer, data_source, preprocessor, masked_lm_prob,
                             mask_token_rate, random_token_rate, input_duplication_factor, min_sequence_len)
    return _Loader


# defaults: bert4rec_ml1m_dataloader.py:8-18 and siblings
BERT4RecML1MDataloader = _flavour("ml_1m", datasets.ML1M, "timestamp", "movie_name", "uid", 200, 40, 0.2, 10, 3)
BERT4RecML20MDataloader = _flavour("ml_20m", datasets.ML20M, "timestamp", "movie_name", "uid", 200, 40, 0.2, 5, 3)
BERT4RecSteamDataloader = _flavour("steam", datasets.Steam, None, "item_id", "user_id", 50, 20, 0.4, 3, 3)
BERT4RecBeautyDataloader = _flavour("beauty", datasets.Beauty, None, "item_id", "user_id", 50, 30, 0.6, 5, 3)
BERT4RecRedditDataloader = _flavour("reddit", datasets.Reddit, None, "item_id", "user_id", 200, 40, 0.2, 2, 3)
for _c, _n in ((BERT4RecML1MDataloader, "BERT4RecML1MDataloader"), (BERT4RecML20MDataloader, "BERT4RecML20MDataloader"),
               (BERT4RecSteamDataloader, "BERT4RecSteamDataloader"), (BERT4RecBeautyDataloader, "BERT4RecBeautyDataloader"),
               (BERT4RecRedditDataloader, "BERT4RecRedditDataloader")):
    _c.__name__ = _c.__qualname__ = _n
