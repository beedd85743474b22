"""Host-side logic of the product (no GPU): batch construction, samplers, tokenizer, metrics, factories and optimizer
bookkeeping, checked against golden vectors captured from the reference (tests/golden/reference_goldens.json), against
the reference's own test expectations, and against the oracle."""
import json
import os
import pathlib

import numpy as np
import pandas as pd
import pytest
import torch

from bert4rec_amd import _lib, config, dataloaders, datasets, evaluation, models, tokenizers, trainers, utils
from bert4rec_amd.dataloaders import dataloader_utils as du
from bert4rec_amd.dataloaders import samplers
from bert4rec_amd.models.components import networks
from bert4rec_amd.trainers import optimizers
from oracle import bert4rec_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_goldens.json")))


@pytest.mark.parametrize("i", range(len(GOLD["apply_dynamic_masking_task"])))
def test_dynamic_masking_matches_reference(i):
    c = GOLD["apply_dynamic_masking_task"][i]
    toks, pos, ids = du.apply_dynamic_masking_task(np.array(c["sequence"], dtype=np.int64), c["max_selections_per_seq"],
                                                   c["mask_token_id"], c["special_token_ids"], c["vocab_size"],
                                                   c["selection_rate"], c["mask_token_rate"], c["random_token_rate"],
                                                   seed=c["seed"])
    assert toks.tolist() == c["masked_token_ids"] and pos.tolist() == c["masked_lm_positions"] and ids.tolist() == c["masked_lm_ids"]
    assert toks.dtype == np.int64 and pos.dtype == np.int64


def test_popularity_ranking_and_split_match_reference():
    g = GOLD["rank_items_by_popularity"]
    assert du.rank_items_by_popularity(list(g["items"])) == g["ranked"]
    s = GOLD["split_sequence_df"]
    tr, va, te = du.split_sequence_df(pd.DataFrame(s["rows"]), "uid", ["item"], s["min_sequence_length"])
    clean = lambda d: [x if isinstance(x, list) else None for x in d["item"].to_list()]
    assert clean(tr) == s["train"] and clean(va) == s["val"] and clean(te) == s["test"]
    with pytest.raises(ValueError):
        du.split_sequence_df(pd.DataFrame(s["rows"]), "nope", ["item"], 3)


def test_samplers_draw_the_reference_samples():
    g = GOLD["samplers"]
    for case in g["cases"]:
        s = samplers.get(case["sampler"], source=list(g["source"]), vocab=list(g["vocab"]), **case["kwargs"])
        assert s.sample() == case["sample"], case
        assert s.sample(without=list(g["without"])) == case["sample_without"], case
        assert not set(case["sample_without"]) & set(g["without"])


def test_sampler_errors_and_factory():
    with pytest.raises(ValueError):
        samplers.get("nope")
    with pytest.raises(ValueError):
        samplers.RandomSampler(vocab=[1, 2, 3], sample_size=5).sample()
    with pytest.raises(ValueError):
        samplers.PopularRandomSampler(sample_size=-1)
    with pytest.raises(ValueError):
        samplers.PopularSampler(sample_size=3).sample()
    s = samplers.PopularSampler(source=[5, 5, 7, 7, 7, 9], sample_size=2)
    assert s.sample() == [7, 5] and s.sample(without=[7]) == [5, 9] and s.is_fully_prepared()


@pytest.mark.parametrize("key", sorted(GOLD["evaluation_metrics"]))
def test_metric_classes_match_reference(key):
    case = GOLD["evaluation_metrics"][key]
    ms = evaluation.default_metrics()
    for r in case["ranks"]:
        for m in ms:
            m.update(r)
    assert {m.name: float(m.result()) for m in ms} == case["results"]
    for m in ms:
        m.reset()
    assert ms[0].result() == 0


def test_tokenizer_roundtrip_and_vocab_file(tmp_path):
    t = tokenizers.get("simple")
    assert [t.tokenize(s) for s in ("[PAD]", "[MASK]", "[UNK]")] == [0, 1, 2]
    assert t.tokenize(["b", "a", "b", b"c"]) == [3, 4, 3, 5] and t.get_vocab_size() == 6
    assert t.detokenize([4, 3, 0], drop_tokens=["[PAD]"]) == ["a", "b"] and t.detokenize(5) == "c"
    assert t.tokenize(pd.Series(["a", "zz"])).tolist() == [4, 6]
    f = tmp_path / "vocab.txt"
    t.export_vocab_to_file(f)
    assert f.read_bytes().split(os.linesep.encode())[3] == b"b|3"
    t2 = tokenizers.SimpleTokenizer(vocab_file_path=f)
    assert t2.get_vocab() == t.get_vocab()
    with pytest.raises(RuntimeError):
        t2.tokenize("never seen")                      # importing a vocab file switches extensibility off
    with pytest.raises(ValueError):
        tokenizers.get("nope")
    with pytest.raises(ValueError):
        tokenizers.SimpleTokenizer().export_vocab_to_file(tmp_path / "e.txt")


def make_loader(**kw):
    ds = datasets.synthetic_dataset(n_users=40, n_items=300, min_len=2, max_len=40, seed=1)
    args = dict(data_source=ds, max_seq_len=20, max_predictions_per_seq=5, input_duplication_factor=2)
    args.update(kw)
    return dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**args)


def test_preprocessor_layout_like_the_reference_tests():
    """tests/datalaoders_tests/preprocessors_tests/bert4rec_preprocessor_tests.py:61-198"""
    dl = make_loader()
    dl.generate_vocab()
    dl._set_preprocessor_properties()
    pp = dl.preprocessor
    seq = list(dl.create_item_list()[:12])
    e = pp.process_element(seq, apply_mlm=False, finetuning=False)
    assert set(e) == {"labels", "input_word_ids", "input_mask"} and all(len(v) == 20 for v in e.values())
    e = pp.process_element(seq, apply_mlm=True, finetuning=False)
    assert len(e) == 6 and all(len(e[k]) == 5 for k in ("masked_lm_ids", "masked_lm_positions", "masked_lm_weights"))
    n = int(e["masked_lm_weights"].sum())
    assert n == max(1, int(12 * 0.2)) and (e["input_word_ids"][e["masked_lm_positions"][:n]] == 1).all()
    assert e["input_mask"].tolist() == [1] * 12 + [0] * 8 and (e["input_word_ids"][12:] == 0).all()
    assert (e["labels"][e["masked_lm_positions"][:n]] == e["masked_lm_ids"][:n]).all()
    f = pp.process_element(seq, apply_mlm=True, finetuning=True)
    assert f["masked_lm_weights"].tolist() == [1, 0, 0, 0, 0] and f["masked_lm_positions"][0] == 11
    assert f["masked_lm_ids"][0] == f["labels"][11] and f["input_word_ids"][11] == 1
    inf = dl.prepare_inference(seq)
    assert inf["input_word_ids"].shape == (1, 20) and inf["input_word_ids"][0, 12] == 1 and inf["masked_lm_positions"][0, 0] == 12
    with pytest.raises(ValueError):
        dl.prepare_inference("not a list")


def test_prepare_training_and_batches():
    dl = make_loader()
    train, val, test = dl.prepare_training(finetuning_split=0.1)
    assert dl.tokenizer.get_vocab_size() == len(set(dl.create_item_list())) + 3
    assert len(train) == 2 * 40 and len(val) == len(test) and len(val) <= 40      # duplication factor 2; short users train-only
    for ex in val.examples + test.examples:
        assert ex["masked_lm_weights"].tolist() == [1, 0, 0, 0, 0]
    n_ft = sum(int(ex["masked_lm_weights"].sum() == 1 and ex["input_word_ids"][ex["input_mask"].sum() - 1] == 1) for ex in train.examples)
    assert n_ft >= int(0.1 * 80)                                                      # the finetuning share masks the last token
    b = dataloaders.make_batches(train, batch_size=32, seed=3)
    assert len(b) == 3 and [x["input_word_ids"].shape[0] for x in b] == [32, 32, 16]  # last batch partial (no drop_remainder)
    assert all(v.dtype == torch.int64 for v in b.batches[0].values()) and set(b.batches[0]) == {
        "labels", "input_word_ids", "input_mask", "masked_lm_ids", "masked_lm_positions", "masked_lm_weights"}
    assert [x["labels"].tolist() for x in b] == [x["labels"].tolist() for x in b]    # cached: identical every epoch
    with pytest.raises(ValueError):
        dl.prepare_training(finetuning_split=1.5)
    with pytest.raises(ValueError):
        make_loader(input_duplication_factor=0)


class _GoldTokenizer:
    """the string -> id table the golden preprocessor cases were captured with (tests/golden/make_reference_goldens.py)"""

    def __init__(self, vocab):
        self.table = {t: i for i, t in enumerate(vocab)}

    def tokenize(self, seq):
        return [self.table[x] for x in seq]

    def get_vocab_size(self):
        return len(self.table)


def _gold_preprocessor():
    from bert4rec_amd.dataloaders.preprocessors import BERT4RecPreprocessor as PP
    g = GOLD["process_element_no_mlm"]
    PP.set_properties(tokenizer=_GoldTokenizer(g["vocab"]), max_seq_len=g["max_seq_len"],
                      max_predictions_per_seq=g["max_predictions_per_seq"], mask_token_id=1, unk_token_id=2, pad_token_id=0,
                      masked_lm_rate=0.2, mask_token_rate=1.0, random_token_rate=0.0)
    return PP, g


def test_truncation_window_and_padding_match_reference_process_element():
    """bert4rec_preprocessor.py:48-72,105-116 run in the build container: last-L truncation, the random window of a long
    training row (same python `random` draws) and the right-padding; the token matrix of the device path holds the same rows."""
    import random
    PP, g = _gold_preprocessor()
    for c in g["cases"]:
        random.seed(c["python_random_seed"])
        e = PP.process_element(list(c["sequence"]), False, c["finetuning"])
        assert set(e) == {"labels", "input_word_ids", "input_mask"}
        for k in ("input_word_ids", "input_mask", "labels"):
            assert e[k].tolist() == c[k] and e[k].dtype == np.int64, (k, c["sequence"])
        random.seed(c["python_random_seed"])
        tm = PP.token_rows(du.SequenceDataset([c["sequence"]]), c["finetuning"])
        assert tm.tokens.tolist() == [c["labels"]] and tm.finetune_rows.tolist() == [int(c["finetuning"])]
    # prepare_inference = the finetuning branch on history[-(L-1):] + ["[UNK]"] (bert4rec_preprocessor.py:125-168)
    hist = [f"item{j}" for j in range(30)]
    inf = PP.prepare_inference(list(hist))
    L = g["max_seq_len"]
    want = PP.tokenizer.tokenize(hist[-(L - 1):])
    assert inf["labels"].tolist() == [want + [2]] and inf["input_word_ids"].tolist() == [want + [1]]
    assert inf["masked_lm_positions"].tolist() == [[L - 1, 0, 0, 0, 0]] and inf["masked_lm_ids"].tolist() == [[2, 0, 0, 0, 0]]
    assert inf["masked_lm_weights"].tolist() == [[1, 0, 0, 0, 0]] and inf["input_mask"].tolist() == [[1] * L]


def test_prepare_inference_matches_the_reference_run():
    """bert4rec_preprocessor.py:125-168 executed in the build container (tests/golden/make_reference_goldens.py section 5): every key,
    value, shape and dtype of what the reference returns for histories shorter than, equal to and longer than max_seq_len - 1; the device path's token matrix
    (finetuning rows of the same histories + [UNK]) holds the same rows."""
    PP, _ = _gold_preprocessor()
    g = GOLD["prepare_inference"]
    for c in g["cases"]:
        got = PP.prepare_inference(list(c["history"]))
        assert set(got) == set(c["out"])
        for k, want in c["out"].items():
            assert np.asarray(got[k]).tolist() == want and str(np.asarray(got[k]).dtype) == c["dtypes"][k], (k, c["history"])
        # the device masker's input for the same request: the [U, L] token matrix row + its finetune flag
        L = g["max_seq_len"]
        tm = PP.token_rows(du.SequenceDataset([c["history"][-(L - 1):] + ["[UNK]"]]), True)
        assert tm.tokens.tolist() == c["out"]["labels"] and tm.finetune_rows.tolist() == [1]
    with pytest.raises(ValueError):
        PP.prepare_inference("item1")


def test_mask_last_token_only_matches_reference():
    for c in GOLD["mask_last_token_only"]:
        toks, pos, ids = du.mask_last_token_only(np.array(c["sequence"], dtype=np.int64), c["mask_token_id"])
        assert toks.tolist() == c["masked_token_ids"] and pos.tolist() == c["masked_lm_positions"]
        assert ids.tolist() == c["masked_lm_ids_by_definition"] and ids.dtype == np.int64


def test_device_masking_datasets_are_token_matrices_with_the_reference_split():
    """prepare_training(device_masking=True): same rows and the same 90 / 10 dynamic / last-token split as the host path
    (bert4rec_dataloader.py:100-108), but unmasked token matrices; batches need the GPU and fail loudly without one."""
    dl = make_loader()
    train, val, test = dl.prepare_training(finetuning_split=0.1, device_masking=True)
    assert isinstance(train, du.TokenMatrixDataset) and train.tokens.shape == (80, 20) and train.tokens.dtype == np.int64
    assert int(train.finetune_rows.sum()) == 8 and val.finetune_rows.all() and test.finetune_rows.all()
    assert train.max_predictions_per_seq == 5 and train.vocab_size == dl.tokenizer.get_vocab_size()
    # same rows as the host path (sequences no longer than max_seq_len: no random window involved)
    short = datasets.synthetic_dataset(n_users=40, n_items=300, min_len=2, max_len=18, seed=1)
    dev_train, _, _ = make_loader(data_source=short).prepare_training(finetuning_split=0.1, device_masking=True)
    host_train, _, _ = make_loader(data_source=short).prepare_training(finetuning_split=0.1)
    assert sorted(map(tuple, dev_train.tokens.tolist())) == sorted(tuple(e["labels"].tolist()) for e in host_train.examples)
    b = dataloaders.make_batches(train, batch_size=32, seed=3, remask_each_epoch=True)
    assert isinstance(b, du.DeviceMaskedBatches) and len(b) == 3
    with pytest.raises(_lib.B4RError):
        next(iter(b.cache_on_device("cpu")))          # no host fallback behind the device masker
    with pytest.raises(ValueError):
        dataloaders.make_batches(host_train, batch_size=32, remask_each_epoch=True)


def test_factories_raise_value_error_on_unknown_ids():
    """tests/trainers_tests/base_trainer_tests.py:21-27, optimizers :15-21, evaluators :21-27"""
    for fn in (lambda: trainers.get("nope", model=None), lambda: optimizers.get("nope"), lambda: evaluation.get("nope"),
               lambda: dataloaders.get_dataloader_factory("nope"), lambda: config.get_encoder_config("nope")):
        with pytest.raises(ValueError):
            fn()


def test_optimizer_schedule_and_decay_rules_match_oracle():
    opt = optimizers.get("adamw")
    hp_o = orc.AdamWConfig()
    for step in (0, 1, 50, 99, 100, 101, 5000, 399999, 400000, 400001):
        assert opt.lr(step) == float(orc.learning_rate(step, hp_o)), step
    names = [n for n, _ in orc.param_names_and_shapes(orc.OracleConfig(vocab_size=50))]
    assert [opt._do_use_weight_decay(n) for n in names] == [orc.uses_weight_decay(n) for n in names]
    custom = optimizers.get("adamw", exclude_from_weight_decay=["bias"])
    enc = networks.Bert4RecEncoder(50, **{**config.get_encoder_config("ml-1m_64"), "max_sequence_length": 16}, device="cpu")
    model = models.BERT4RecModel(enc)
    with pytest.raises(ValueError):
        model.compile(optimizer=custom)            # LayerNorm variables decay: a per-element mask, which lives on the GPU
    model.compile()
    assert model._hp.clip_norm == 5.0 and model._hp.num_warmup_steps == 100 and abs(model._hp.epsilon - 1e-6) < 1e-12


def test_encoder_surface_and_loud_failures():
    cfgd = {**config.get_encoder_config("ml-1m_64"), "max_sequence_length": 16}
    enc = networks.Bert4RecEncoder(50, device="cpu", **cfgd)
    assert enc.get_config()["vocab_size"] == 50 and enc.get_embedding_table().shape == (50, 64)
    legacy = networks.Bert4RecEncoder(50, hidden_size=64, num_layers=1, num_attention_heads=2, max_sequence_length=16,
                                      intermediate_size=128, dropout_rate=0.3, attention_dropout_rate=0.1, device="cpu")
    c = legacy.get_config()
    assert (c["inner_dim"], c["output_dropout"], c["attention_dropout"]) == (128, 0.3, 0.1)   # bert4rec_encoder.py:82-93
    with pytest.raises(ValueError):
        networks.Bert4RecEncoder(50, hidden_size=96, num_attention_heads=3, device="cpu")     # head_dim must be 32
    with pytest.raises(NotImplementedError):
        networks.Bert4RecEncoder(50, hidden_size=64, num_attention_heads=2, norm_first=True, device="cpu")
    model = models.BERT4RecModel(enc)
    batch = {k: v for k, v in orc.synthetic_batch(2, 16, 4, 50).items()}
    with pytest.raises(_lib.B4RError):
        model(batch)                               # no CPU fallback: computing without the GPU fails loudly
    with pytest.raises(RuntimeError):
        model.train_step(batch)                    # not compiled
    names = model.trainable_variables
    assert "pooler_transform/kernel" not in names and "cls/predictions/output_bias/bias" in names
    w = model.get_weights()
    assert w["transformer/layer_0/self_attention/query/kernel"].shape == (64, 2, 32)
    assert w["transformer/layer_1/self_attention/attention_output/kernel"].shape == (2, 32, 64)
    assert float(w["embeddings/layer_norm/gamma"].min()) == 1.0 and float(w["cls/predictions/output_bias/bias"].abs().max()) == 0.0


def test_weights_roundtrip_through_safetensors_and_wrapper_rules(tmp_path):
    cfgd = {**config.get_encoder_config("steam_64"), "max_sequence_length": 16}
    m1 = models.BERT4RecModel(networks.Bert4RecEncoder(40, device="cpu", seed=5, **cfgd))
    m2 = models.BERT4RecModel(networks.Bert4RecEncoder(40, device="cpu", seed=6, **cfgd))
    f = tmp_path / "w.safetensors"
    m1.save_weights(f)
    m2.load_weights(f)
    assert torch.equal(m1.engine.params, m2.engine.params) and torch.equal(m1.engine.pooler, m2.engine.pooler)
    wrapper = models.BERT4RecModelWrapper(m1)
    assert wrapper.get_meta_config()["model"] == "BERT4Rec" and wrapper.get_meta_config()["encoder_config"]["hidden_size"] == 64
    with pytest.raises(RuntimeError):
        wrapper.save(tmp_path / "m", mode=2)       # bert4rec_wrapper.py:60-68: not compiled / no train step yet
    assert models.model_utils.determine_model_path(pathlib.Path("x"), 2) == pathlib.Path("x")
    with pytest.raises(ValueError):
        models.model_utils.determine_model_path(pathlib.Path("x"), 7)


def test_config_table_and_json_loader(tmp_path):
    c = config.get_encoder_config("ml-20m_256.json")
    assert c == {"attention_dropout": 0.1, "output_dropout": 0.1, "hidden_size": 256, "inner_dim": 1024,
                 "max_sequence_length": 200, "num_attention_heads": 8, "num_layers": 2}
    files = config.write_config_files(tmp_path)
    assert len(files) == 13 and utils.load_json_config(tmp_path / "ml-1m_64.json") == config.get_encoder_config("ml-1m_64")
    with pytest.raises(ValueError):
        utils.load_json_config(tmp_path / "missing.json")
    assert all(config.get_encoder_config(n)["hidden_size"] == 32 * config.get_encoder_config(n)["num_attention_heads"]
               for n in config.available_configs())


def test_evaluator_candidate_sampling_contract():
    """bert4rec_evaluator.py:91-101: 100 negatives that avoid the user's items, ground truth appended last."""
    dl = make_loader()
    _, _, test = dl.prepare_training()
    ev = evaluation.get(dataloader=dl)
    batch = dataloaders.make_batches(test, batch_size=8, seed=0).batches[0]
    cand, gt = ev.sample_candidates(batch)
    assert cand.shape == (8, 101) and (cand[:, 100] == gt).all()
    for r in range(8):
        seen = set(batch["labels"][r].tolist())
        assert not (set(cand[r, :100].tolist()) & seen) and len(set(cand[r].tolist())) == 101
    ev2 = evaluation.get(sampler=samplers.get("random", sample_size=100))
    with pytest.raises(ValueError):
        ev2.evaluate(model=None, test_data=[])


def test_length_bucketing_and_padding_trim_keep_the_rows_and_cut_only_padding():
    """make_batches(bucket_by_length=W, trim_padding=True): not in the reference (it pads every row to max_seq_len,
    bert4rec_preprocessor.py:105-110).  Same rows as without, every batch cut to a multiple of 16 columns that still holds its
    longest sequence and every masked position; the cut columns are padding only."""
    assert du.sequence_lengths(np.array([[5, 6, 0, 0], [0, 0, 0, 0], [1, 2, 3, 4], [7, 0, 0, 0]])).tolist() == [2, 0, 4, 1]
    assert [du.trimmed_length(n, 200) for n in (0, 1, 16, 17, 199, 200)] == [16, 16, 16, 32, 200, 200]
    order = du.bucket_order(np.arange(12), np.array([5, 1, 9, 3, 7, 2, 8, 4, 6, 10, 11, 0]), 2, 3, seed=0)
    assert sorted(order.tolist()) == list(range(12)) and sorted(order[:6].tolist()) == [0, 1, 2, 3, 4, 5]   # windows keep their rows
    assert (du.bucket_order(np.arange(12), np.arange(12), 2, 0, seed=0) == np.arange(12)).all()
    ds = datasets.synthetic_dataset(n_users=60, n_items=300, min_len=2, max_len=70, seed=2)
    dl = make_loader(data_source=ds, max_seq_len=64, max_predictions_per_seq=8, input_duplication_factor=1)
    train, _, _ = dl.prepare_training(finetuning_split=0.1)
    plain = dataloaders.make_batches(train, batch_size=16, seed=3)
    cut = dataloaders.make_batches(train, batch_size=16, seed=3, bucket_by_length=4, trim_padding=True)
    rows = lambda bs: sorted(tuple(r[r != 0].tolist()) for b in bs for r in b["labels"].numpy())
    assert rows(plain) == rows(cut) and len(plain) == len(cut)
    widths = [b["input_word_ids"].shape[1] for b in cut]
    assert all(w % 16 == 0 and 16 <= w <= 64 for w in widths) and min(widths) < 64
    for b in cut:
        w = b["input_word_ids"].shape[1]
        assert b["input_mask"].shape[1] == w and b["labels"].shape[1] == w
        p = b["masked_lm_positions"].shape[1]
        assert p in (4, 8) and b["masked_lm_ids"].shape[1] == p and b["masked_lm_weights"].shape[1] == p
        assert int(b["masked_lm_weights"].sum(1).max()) > p - 4        # the slots that are cut are padded slots
        assert int(b["input_mask"].sum(1).max()) > w - 16 or w == 16          # no narrower multiple of 16 would do
        assert int(b["masked_lm_positions"].max()) < w
    dev_train, _, _ = dl.prepare_training(finetuning_split=0.1, device_masking=True)
    dev = dataloaders.make_batches(dev_train, batch_size=16, seed=3, bucket_by_length=4, trim_padding=True)
    lens = du.sequence_lengths(dev_train.tokens)
    assert sorted(dev.order.tolist()) == list(range(len(dev_train)))
    assert dev.batch_columns == [du.trimmed_length(lens[dev.order[s:s + 16]].max(), 64) for s in range(0, len(dev_train), 16)]
    assert all(p in (4, 8) for p in dev.batch_slots) and min(dev.batch_slots) == 4


def test_data_parallel_fit_consumes_every_batch_and_pads_the_last_round():
    """BERT4RecModel.fit with an initialised process group: rank r trains on batches r, r + world, ...; every rank makes the same
    number of rounds (= all-reduces) and NO batch is dropped -- where the last round has no batch for a rank it yields None (the
    rank then contributes zeros, Engine.dp_idle_step)."""
    from bert4rec_amd.models.bert4rec_model import dp_shard
    data = list(range(11))
    assert list(dp_shard(data, 0, 1)) == data
    got = [list(dp_shard(data, r, 4)) for r in range(4)]
    assert got == [[0, 4, 8], [1, 5, 9], [2, 6, 10], [3, 7, None]]
    assert all(len(g) == 3 for g in got)
    assert sorted(b for g in got for b in g if b is not None) == data           # every batch exactly once
    assert [list(dp_shard(list(range(8)), r, 4)) for r in range(4)] == [[0, 4], [1, 5], [2, 6], [3, 7]]   # no padding needed
    assert [list(dp_shard([0], r, 2)) for r in range(2)] == [[0], [None]]


def test_hash_uniform_is_strictly_inside_the_unit_interval():
    """The uniform the device masker and the negative sampler draw from a hash word (b4r_uniform23, host/device inline in
    csrc/b4r_common.h, probed through b4r_uniform_from_hash): python's random.random() is in [0, 1) (dataloader_utils.py:245-253),
    so with mask_token_rate = 1.0 `u < rate` must hold for EVERY word.  Round 3's 24-bit form gave exactly 1.0f for the words
    0xFFFFFF00..0xFFFFFFFF: one selected position in 2^24 kept its real token while being a prediction target (a label leak)."""
    lib = _lib.load()
    words = [0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFF00, 0xFFFFFFFE, 0xFFFFFFFF]
    words += [int(w) for w in np.random.default_rng(0).integers(0, 2 ** 32, size=2000, dtype=np.uint64)]
    for w in words:
        u = lib.b4r_uniform_from_hash(w)
        assert 0.0 < u < 1.0, hex(w)
        assert u == (float(w >> 9) + 0.5) / 8388608.0, hex(w)          # exact in fp32: no rounding anywhere
    assert lib.b4r_uniform_from_hash(0xFFFFFFFF) == 1.0 - 2.0 ** -24    # the largest value; the 24-bit form rounded to 1.0f here
    assert np.float32((np.float32(0xFFFFFF) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)) == np.float32(1.0)   # ... as this shows
