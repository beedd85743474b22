#!/usr/bin/env python3
"""Generates tests/golden/reference_goldens.json by RUNNING the numpy/python parts of the reference
(/root/reference, read-only) in the build container.  Only data (inputs and expected outputs) is written; no reference
source travels.  The GPU box never runs this script (it has no /root/reference).

What can be executed here: TensorFlow, tensorflow_models and absl are not installed, so the float path cannot run.
The integer/numpy functions below never touch TensorFlow; to import their modules the script registers inert
placeholder modules for the missing imports (they are never called by the captured functions).

    python tests/golden/make_reference_goldens.py
"""
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_goldens.json")


class _Inert(types.ModuleType):
    """Module placeholder: any attribute is another placeholder; calling it returns a placeholder."""

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        child = _Inert(self.__name__ + "." + name)
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        return _Inert(self.__name__ + "()")

    def __mro_entries__(self, bases):
        return (object,)

    def __or__(self, other):
        return self

    def __getitem__(self, item):
        return self


def install_placeholders():
    for name in ("absl", "absl.logging", "tensorflow", "tensorflow.python", "tensorflow.python.framework",
                 "tensorflow.python.framework.ops", "tensorflow_models", "wget", "zstandard"):
        if name not in sys.modules:
            sys.modules[name] = _Inert(name)
    sys.modules["absl"].logging = sys.modules["absl.logging"]


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    os.environ.setdefault("VIRTUAL_ENV", "/tmp")
    install_placeholders()
    sys.path.insert(0, REF)
    out = {"generated_from": "maneymarkus/BERT4Rec @ /root/reference (numpy/python parts only)"}

    # ---- 1. evaluation metrics (no placeholders involved) -----------------------------------------------------------
    em = load_by_path("ref_evaluation_metrics", "bert4rec/evaluation/evaluation_metrics.py")
    rank_lists = {"ranks_1": [1, 2, 3, 4, 5], "ranks_2": [1, 5, 10, 15, 20], "ranks_3": [2, 8, 4, 13, 20, 6, 3, 11, 2, 5],
                  "ranks_4": [1], "ranks_5": [101, 57, 1, 10, 11, 5, 6, 2]}
    metrics_out = {}
    for key, ranks in rank_lists.items():
        ms = [em.Counter(name="Valid Ranks"), em.NDCG(1), em.NDCG(5), em.NDCG(10), em.HR(1), em.HR(5), em.HR(10), em.MAP()]
        for r in ranks:
            for m in ms:
                m.update(r)
        metrics_out[key] = {"ranks": ranks, "results": {m.name: float(m.result()) for m in ms}}
    out["evaluation_metrics"] = metrics_out

    # ---- 2. masking / popularity / split utilities --------------------------------------------------------------------
    du = load_by_path("ref_dataloader_utils", "bert4rec/dataloaders/dataloader_utils.py")
    cases = []
    for seed, n, max_sel, rate, mtr, rtr, vocab in [(7, 20, 40, 0.2, 1.0, 0.0, 3709), (1, 200, 40, 0.2, 1.0, 0.0, 3709),
                                                    (2, 50, 20, 0.4, 1.0, 0.0, 13047), (3, 7, 40, 0.2, 1.0, 0.0, 100),
                                                    (4, 3, 40, 0.2, 1.0, 0.0, 100), (5, 60, 5, 0.6, 1.0, 0.0, 500),
                                                    (6, 30, 40, 0.5, 0.8, 0.1, 50), (8, 30, 40, 0.5, 0.5, 0.5, 50)]:
        rng = np.random.default_rng(seed)
        seq = rng.integers(3, vocab, size=n).astype(np.int64)
        toks, pos, ids = du.apply_dynamic_masking_task(seq.copy(), max_sel, 1, [2, 0], vocab, rate, mtr, rtr, seed=seed)
        cases.append({"seed": seed, "sequence": seq.tolist(), "max_selections_per_seq": max_sel, "mask_token_id": 1,
                      "special_token_ids": [2, 0], "vocab_size": vocab, "selection_rate": rate, "mask_token_rate": mtr,
                      "random_token_rate": rtr, "masked_token_ids": np.asarray(toks).tolist(),
                      "masked_lm_positions": np.asarray(pos).tolist(), "masked_lm_ids": np.asarray(ids).tolist()})
    out["apply_dynamic_masking_task"] = cases
    pop_in = ["a", "b", "a", "c", "b", "a", "d", "d", "e", "d", "d"]
    out["rank_items_by_popularity"] = {"items": pop_in, "ranked": du.rank_items_by_popularity(list(pop_in))}
    rows = []
    for uid, n in [(1, 6), (2, 2), (3, 3), (4, 9), (5, 1)]:
        for j in range(n):
            rows.append({"uid": uid, "item": f"i{uid}_{j}", "timestamp": 100 * uid + j})
    df = pd.DataFrame(rows)
    tr, va, te = du.split_sequence_df(df, "uid", ["item"], 3)
    clean = lambda d: [x if isinstance(x, list) else None for x in d["item"].to_list()] if len(d.columns) else []
    out["split_sequence_df"] = {"rows": rows, "min_sequence_length": 3, "train": clean(tr), "val": clean(va), "test": clean(te)}

    # ---- 3. samplers (seeded) --------------------------------------------------------------------------------------------
    smp = importlib.import_module("bert4rec.dataloaders.samplers")
    rng = np.random.default_rng(0)
    vocab = list(range(0, 60))
    source = rng.choice(np.arange(3, 60), size=800, p=(lambda p: p / p.sum())(1.0 / np.arange(1, 58))).tolist()
    without = [3, 4, 5, 9, 0]
    s_out = {"vocab": vocab, "source": source, "without": without, "cases": []}
    for ident, kw in [("pop_random", dict(sample_size=10, seed=11)), ("pop_random", dict(sample_size=25, seed=12)),
                      ("random", dict(sample_size=10, seed=13)), ("popular", dict(sample_size=10))]:
        s = smp.get(ident, source=list(source), vocab=list(vocab), **kw)
        s_out["cases"].append({"sampler": ident, "kwargs": kw, "sample": s.sample(),
                               "sample_without": s.sample(without=list(without))})
    out["samplers"] = s_out

    # ---- 4. preprocessor: truncation / random window / padding, last-token mask ------------------------------------------
    # process_element never touches TensorFlow when apply_mlm is False (bert4rec_preprocessor.py:48-72,105-116): token lookup,
    # last-L truncation (finetuning or short rows), the random window of a long training row (python `random`, seeded here)
    # and the right-padding of input_word_ids / input_mask / labels.  With apply_mlm the training branch reseeds `random`
    # from the OS (seed=None) and the finetuning branch wraps the label in tf.constant (a placeholder here), so for
    # mask_last_token_only (dataloader_utils.py:264-269) only its two numpy outputs are captured here; prepare_inference
    # (bert4rec_preprocessor.py:125-168) is captured directly in section 5.
    import random as _random
    pre_mod = importlib.import_module("bert4rec.dataloaders.preprocessors.bert4rec_preprocessor")
    vocab_items = [f"item{j}" for j in range(40)]

    class _Lookup:
        """the tokenizer is an INPUT of process_element (the reference's SimpleTokenizer needs tf types for list input): a plain
        string -> id table with the dataloader's id convention PAD 0, MASK 1, UNK 2, items from 3"""
        table = {t: i for i, t in enumerate(["[PAD]", "[MASK]", "[UNK]"] + vocab_items)}

        def tokenize(self, seq):
            return [self.table[x] for x in seq]

        def get_vocab_size(self):
            return len(self.table)

    tok = _Lookup()
    PP = pre_mod.BERT4RecPreprocessor
    L_, P_ = 12, 5
    PP.set_properties(tokenizer=tok, max_seq_len=L_, max_predictions_per_seq=P_, mask_token_id=1, unk_token_id=2, pad_token_id=0,
                      masked_lm_rate=0.2, mask_token_rate=1.0, random_token_rate=0.0)
    pe_cases = []
    for n, finetune, seed in [(3, False, 1), (3, True, 1), (12, False, 2), (12, True, 2), (13, True, 3), (30, True, 4), (30, False, 5),
                              (30, False, 6), (17, False, 7), (1, True, 8)]:
        seq = [vocab_items[(7 * j + n) % 40] for j in range(n)]
        _random.seed(seed)
        r = PP.process_element(list(seq), False, finetune)
        pe_cases.append({"sequence": seq, "finetuning": finetune, "python_random_seed": seed,
                         "input_word_ids": np.asarray(r["input_word_ids"]).tolist(), "input_mask": np.asarray(r["input_mask"]).tolist(),
                         "labels": np.asarray(r["labels"]).tolist()})
    out["process_element_no_mlm"] = {"max_seq_len": L_, "max_predictions_per_seq": P_, "vocab": ["[PAD]", "[MASK]", "[UNK]"] + vocab_items,
                                     "cases": pe_cases}
    ml_cases = []
    for n in (1, 2, 7, 12):
        seq = np.arange(3, 3 + n, dtype=np.int64)
        toks, pos, _ = du.mask_last_token_only(seq.copy(), 1)
        ml_cases.append({"sequence": seq.tolist(), "mask_token_id": 1, "masked_token_ids": np.asarray(toks).tolist(),
                         "masked_lm_positions": np.asarray(pos).tolist(), "masked_lm_ids_by_definition": [int(seq[-1])]})
    out["mask_last_token_only"] = ml_cases

    # ---- 5. prepare_inference, captured directly (bert4rec_preprocessor.py:125-168) -----------------------------------------
    # It calls process_element(history[-(L-1):] + ["[UNK]"], apply_mlm=True, finetuning=True) and wraps every array with
    # tf.expand_dims(tf.constant(value), axis=0); the finetuning branch of process_element wraps its label in tf.constant.  Both tf
    # calls are given their numpy meaning for this capture (constant = asarray, expand_dims = numpy's): the reference's own python /
    # numpy code then runs end to end and every key it returns is recorded.
    tf_stub = pre_mod.tf
    tf_stub.constant = lambda v, *a, **k: np.asarray(v)
    tf_stub.expand_dims = lambda v, axis=0: np.expand_dims(np.asarray(v), axis)
    du_tf = getattr(du, "tf", None)
    if du_tf is not None:
        du_tf.constant = tf_stub.constant
    pi_cases = []
    for n in (1, 4, L_ - 1, L_, 30):
        hist = [vocab_items[(5 * j + n) % 40] for j in range(n)]
        r = PP.prepare_inference(list(hist))
        pi_cases.append({"history": hist, "out": {k: np.asarray(v).tolist() for k, v in r.items()},
                         "dtypes": {k: str(np.asarray(v).dtype) for k, v in r.items()}})
    out["prepare_inference"] = {"max_seq_len": L_, "max_predictions_per_seq": P_, "cases": pi_cases}

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
