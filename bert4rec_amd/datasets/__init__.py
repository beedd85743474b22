"""Interaction-log readers (mirror bert4rec/datasets/*).  There is no network in the build/GPU environment, so nothing is
downloaded: each reader parses the same raw files the reference downloads if they are present locally
(default <project root>/datasets/<name>/, override with the B4R_DATA_DIR environment variable) and raises otherwise.
SyntheticInteractions generates a Zipf-popularity log with the same columns for tests and examples."""
import os
import pathlib
from typing import Optional

import numpy as np
import pandas as pd


def data_root() -> pathlib.Path:
    env = os.environ.get("B4R_DATA_DIR")
    if env:
        return pathlib.Path(env)
    return pathlib.Path(__file__).resolve().parent.parent.parent / "datasets"


class BaseDataset:
    load_n_records: Optional[int] = None
    name = "base"

    @classmethod
    def dest(cls) -> pathlib.Path:
        return data_root() / cls.name

    @classmethod
    def is_available(cls) -> bool:
        return cls.dest().exists()

    @classmethod
    def load_data(cls) -> pd.DataFrame:
        if not cls.is_available():
            raise FileNotFoundError(f"dataset '{cls.name}' not found at {cls.dest()} (no network: place the raw files "
                                    f"there or set B4R_DATA_DIR)")
        return cls.extract_data()

    @classmethod
    def set_load_n_records(cls, n_records: int):
        cls.load_n_records = n_records
        return cls

    @classmethod
    def extract_data(cls) -> pd.DataFrame:
        raise NotImplementedError


class ML1M(BaseDataset):
    """ml-1m/ratings.dat + movies.dat -> columns uid, sid, rating, timestamp, movie_name, categories (ml_1m.py:38-57)"""
    name = "ml-1m"

    @classmethod
    def extract_data(cls) -> pd.DataFrame:
        df = pd.read_csv(cls.dest() / "ratings.dat", sep="::", header=None, engine="python", encoding="iso-8859-1",
                         nrows=cls.load_n_records)
        df.columns = ["uid", "sid", "rating", "timestamp"]
        movies = pd.read_csv(cls.dest() / "movies.dat", sep="::", header=None, engine="python", encoding="iso-8859-1",
                             nrows=cls.load_n_records)
        movies.columns = ["sid", "movie_name", "categories"]
        return pd.merge(df, movies)


class ML20M(BaseDataset):
    name = "ml-20m"

    @classmethod
    def extract_data(cls) -> pd.DataFrame:
        df = pd.read_csv(cls.dest() / "ratings.csv", nrows=cls.load_n_records)
        df.columns = ["uid", "sid", "rating", "timestamp"]
        movies = pd.read_csv(cls.dest() / "movies.csv")
        movies.columns = ["sid", "movie_name", "categories"]
        return pd.merge(df, movies)


class _PairFile(BaseDataset):
    """`user item` per line, already in chronological order per user (steam.py:35-52 and the beauty/reddit readers)."""
    file_name = "data.txt"

    @classmethod
    def is_available(cls) -> bool:
        return (cls.dest() / cls.file_name).is_file()

    @classmethod
    def extract_data(cls) -> pd.DataFrame:
        users, items = [], []
        with open(cls.dest() / cls.file_name, "rb") as f:
            for i, line in enumerate(f):
                if cls.load_n_records is not None and i >= cls.load_n_records:
                    break
                parts = line.split()
                users.append(int(parts[0]))
                items.append(parts[1].decode())
        return pd.DataFrame({"user_id": users, "item_id": items})


class Steam(_PairFile):
    name = "steam"
    file_name = "steam.txt"


class Beauty(_PairFile):
    name = "beauty"
    file_name = "beauty.txt"


class Reddit(_PairFile):
    name = "reddit"
    file_name = "reddit.txt"


def make_synthetic(columns=("uid", "movie_name", "timestamp"), n_users: int = 200, n_items: int = 300,
                   min_len: int = 5, max_len: int = 60, zipf_a: float = 1.2, seed: int = 0, order: float = 0.0) -> pd.DataFrame:
    """Zipf-popularity interaction log; column names follow the requested dataset flavour.
    order > 0 adds something a sequence model can learn: with that probability the next item is the fixed successor of the
    previous one (a random permutation of the catalogue), else a popularity draw -- so NDCG of a trained model rises well above
    the popularity baseline, which makes the number a check of the whole train + evaluate pipeline."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, n_items + 1) ** zipf_a
    p /= p.sum()
    succ = rng.permutation(n_items)
    rows = []
    t = 0
    for u in range(1, n_users + 1):
        n = int(rng.integers(min_len, max_len + 1))
        if order > 0:
            draws = rng.choice(n_items, size=n, p=p)
            follow = rng.random(n) < order
            seq = np.empty(n, dtype=np.int64)
            seq[0] = draws[0]
            for j in range(1, n):
                seq[j] = succ[seq[j - 1]] if follow[j] else draws[j]
        else:
            seq = rng.choice(n_items, size=min(n, n_items), replace=False, p=p)
        for it in seq:
            t += 1
            rows.append((u, f"item_{int(it):05d}", t))
    df = pd.DataFrame(rows, columns=["_u", "_i", "_t"])
    user_col, item_col = columns[0], columns[1]
    out = pd.DataFrame({user_col: df["_u"], item_col: df["_i"]})
    if len(columns) > 2:
        out[columns[2]] = df["_t"]
    return out


def synthetic_dataset(columns=("uid", "movie_name", "timestamp"), **kw):
    """A BaseDataset subclass (the dataloaders take a class as data_source) over make_synthetic(**kw)."""
    frame = make_synthetic(columns, **kw)

    class SyntheticInteractions(BaseDataset):
        name = "synthetic"

        @classmethod
        def is_available(cls):
            return True

        @classmethod
        def extract_data(cls):
            return frame.copy()

    return SyntheticInteractions
