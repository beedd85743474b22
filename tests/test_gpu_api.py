"""GPU tests of the reference-shaped API (models / trainers / evaluation / apps) on the HIP path."""
import numpy as np
import pytest
import torch

from bert4rec_amd import config, dataloaders, datasets, evaluation, models, trainers
from bert4rec_amd.apps import Ranker, Recommender
from bert4rec_amd.models.components import networks
from bert4rec_amd.trainers import optimizers, trainer_utils
from oracle import bert4rec_oracle as orc

pytestmark = pytest.mark.gpu


def make_loader(**kw):
    ds = datasets.synthetic_dataset(n_users=120, n_items=300, min_len=4, max_len=40, seed=1)
    args = dict(data_source=ds, max_seq_len=24, max_predictions_per_seq=6, input_duplication_factor=2)
    args.update(kw)
    return dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**args)


def make_model(vocab, L=24, dropout=0.0, seed=3):
    cfgd = {**config.get_encoder_config("ml-1m_64"), "max_sequence_length": L, "output_dropout": dropout,
            "attention_dropout": dropout}
    return models.BERT4RecModel(networks.Bert4RecEncoder(vocab, seed=seed, **cfgd))


def oracle_of(model):
    c = model.encoder.get_config()
    cfg_o = orc.OracleConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], num_layers=c["num_layers"],
                             num_attention_heads=c["num_attention_heads"], max_sequence_length=c["max_sequence_length"],
                             inner_dim=c["inner_dim"])
    return cfg_o, {k: v.clone() for k, v in model.get_weights().items()}


def test_model_call_keys_shapes_and_values():
    """tests/models_tests/bert4rec_model_tests.py:42-95 (keys, shapes) + values against the oracle"""
    model = make_model(50)
    batch = orc.synthetic_batch(5, 24, 6, 50, seed=1, ragged=True)
    out = model(batch, training=False)
    assert set(out) == {"sequence_output", "pooled_output", "encoder_outputs", "mlm_logits"}
    assert out["sequence_output"].shape == (5, 24, 64) and out["pooled_output"].shape == (5, 64)
    assert len(out["encoder_outputs"]) == 2 and out["mlm_logits"].shape == (5, 6, 50)
    cfg_o, params = oracle_of(model)
    ref = orc.model_forward(params, batch, cfg_o)
    assert float((out["mlm_logits"].cpu() - ref["mlm_logits"]).abs().max()) < 1e-3
    assert float((out["pooled_output"].cpu() - ref["pooled_output"]).abs().max()) < 1e-3
    enc_only = model({k: batch[k] for k in ("input_word_ids", "input_mask")})
    assert "mlm_logits" not in enc_only
    enc = model.encoder({"input_word_ids": batch["input_word_ids"], "input_mask": batch["input_mask"]})
    assert set(enc) == {"sequence_output", "pooled_output", "encoder_outputs"}
    with pytest.raises(ValueError):
        model.encoder([batch["input_word_ids"]])


def test_loss_and_metric_callables_match_oracle():
    logits = torch.randn(4, 6, 50)
    y = torch.randint(0, 50, (4, 6))
    y[:, -2:] = 0
    loss = trainer_utils.MaskedSparseCategoricalCrossentropy()(y.cuda(), logits.cuda())
    assert abs(float(loss) - float(orc.masked_sparse_categorical_crossentropy(y, logits))) < 1e-5
    assert abs(float(trainer_utils.masked_accuracy(y.cuda(), logits.cuda())) - float(orc.masked_accuracy(y, logits))) < 1e-7
    assert abs(float(trainer_utils.sparse_categorical_accuracy(y.cuda(), logits.cuda()))
               - float(orc.sparse_categorical_accuracy(y, logits))) < 1e-7


def test_test_step_and_evaluate_values_follow_the_keras_averaging_rules():
    """bert4rec_model.py:175-192 (test_step) and Keras' evaluate() on top of it, VALUES against the oracle on two batches of
    different size: `loss` = batch-size-weighted mean of the per-batch masked CE (compiled_loss keeps a Mean with sample_weight =
    batch size), `sparse_categorical_accuracy` = matches / slots over everything seen (MeanMetricWrapper over elements, PAD slots
    included: bert4rec_trainer.py:28-33), `masked_accuracy` = UNWEIGHTED mean of the per-batch scalars (a function metric,
    trainer_utils.py:49-60).  test_step returns the running values, evaluate() the same with a prefix."""
    model = make_model(60)
    trainers.get(model=model).initialize_model()
    # argmax matches that are neither all nor none: item 5 wins every argmax (output bias), and the labels of the valid slots are drawn
    # from {5, 6, 7} in the first batch and {5, 6} in the second -> masked accuracy about 1/3 and 1/2
    w = model.get_weights()
    w["cls/predictions/output_bias/bias"][5] = 6.0
    model.set_weights(w)
    b1 = orc.synthetic_batch(7, 24, 6, 60, seed=11, ragged=True)
    b2 = orc.synthetic_batch(3, 24, 6, 60, seed=12, ragged=True)
    for b, n in ((b1, 3), (b2, 2)):
        ids = b["masked_lm_ids"]
        valid = ids != 0
        ids[valid] = 5 + torch.arange(int(valid.sum())) % n
    cfg_o, params = oracle_of(model)
    per = []
    for b in (b1, b2):
        lg = orc.model_forward(params, b, cfg_o)["mlm_logits"]
        y = b["masked_lm_ids"]
        per.append(dict(n=y.shape[0], loss=float(orc.masked_sparse_categorical_crossentropy(y, lg)),
                        macc=float(orc.masked_accuracy(y, lg)), hits=float((y == lg.argmax(2)).sum()), slots=float(y.numel())))
    want1 = {"loss": per[0]["loss"], "sparse_categorical_accuracy": per[0]["hits"] / per[0]["slots"], "masked_accuracy": per[0]["macc"]}
    want2 = {"loss": (per[0]["loss"] * 7 + per[1]["loss"] * 3) / 10,
             "sparse_categorical_accuracy": (per[0]["hits"] + per[1]["hits"]) / (per[0]["slots"] + per[1]["slots"]),
             "masked_accuracy": (per[0]["macc"] + per[1]["macc"]) / 2}
    assert abs(per[0]["loss"] - per[1]["loss"]) > 1e-3 and per[0]["macc"] != per[1]["macc"]     # the rules can be told apart
    assert abs(want2["loss"] - (per[0]["loss"] + per[1]["loss"]) / 2) > 1e-4
    model.reset_metrics()
    got1 = model.test_step(b1)
    got2 = model.test_step(b2)
    ev = model.evaluate([b1, b2], prefix="val_")
    for got, want in ((got1, want1), (got2, want2), ({k[4:]: v for k, v in ev.items()}, want2)):
        assert set(got) == set(want)
        assert abs(got["loss"] - want["loss"]) < 1e-4, (got, want)                       # fp32 sums of ~40 terms of size ~4
        assert abs(got["sparse_categorical_accuracy"] - want["sparse_categorical_accuracy"]) < 1e-6, (got, want)
        assert abs(got["masked_accuracy"] - want["masked_accuracy"]) < 1e-6, (got, want)
    assert set(ev) == {"val_loss", "val_sparse_categorical_accuracy", "val_masked_accuracy"}
    assert model.evaluate([b1, b2], steps=1)["loss"] == pytest.approx(want1["loss"], abs=1e-4)   # validation_steps


def test_train_and_evaluate_lifecycle(tmp_path):
    """the call sequence of examples/bert4rec_ml_1m_example.py:14-91 on a synthetic log"""
    dl = make_loader()
    train, val, test = dl.prepare_training()
    model = make_model(dl.tokenizer.get_vocab_size(), dropout=0.1)
    trainer = trainers.get(model=model)
    trainer.initialize_model(optimizer=optimizers.get("adamw", init_lr=2e-3, num_warmup_steps=5, num_train_steps=2000))
    tb = dataloaders.make_batches(train, batch_size=64, seed=1)
    vb = dataloaders.make_batches(val, batch_size=64, seed=1)
    ckpt = tmp_path / "ckpt" / "best"
    trainer.append_callback(trainers.EarlyStopping(monitor="val_loss", patience=50))
    hist = trainer.train(tb, vb, checkpoint_path=ckpt, epochs=6)
    h = hist.history
    assert set(h) == {"loss", "sparse_categorical_accuracy", "masked_accuracy", "val_loss", "val_sparse_categorical_accuracy",
                      "val_masked_accuracy"}
    assert len(h["loss"]) == 6 and h["loss"][-1] < h["loss"][0] - 0.05, h["loss"]
    assert all(np.isfinite(v).all() for v in h.values())
    assert (tmp_path / "ckpt" / "best.safetensors").is_file()          # best-val_masked_accuracy weights-only checkpoint
    assert model.engine.read_state()["step"] == 6 * len(tb)
    # evaluator: reference integration test tests/evaluators_tests/bert4rec_evaluator_tests.py:36-71
    evaluator = evaluation.get(dataloader=dl)
    testb = dataloaders.make_batches(test, batch_size=64, seed=1)
    evaluator.evaluate(model, testb)
    res = evaluator.get_metrics_results()
    assert list(res) == ["Valid Ranks", "NDCG@1", "NDCG@5", "NDCG@10", "HR@1", "HR@5", "HR@10", "MAP"]
    assert res["Valid Ranks"] == len(test) and all(0 <= v <= 1 for k, v in res.items() if k != "Valid Ranks")
    assert res["HR@10"] > 0.12          # better than the 10/101 chance level after a few epochs on Zipf data
    p = evaluator.save_results(tmp_path)
    assert p.name == "eval_results.json" and p.is_file()
    # save / load (bert4rec_wrapper.py:46-124)
    wrapper = models.BERT4RecModelWrapper(model)
    trainer.update_wrapper_meta_info(wrapper, dl)
    wrapper.save(tmp_path / "model", dl.get_tokenizer(), mode=2)
    loaded = models.BERT4RecModelWrapper.load(tmp_path / "model", mode=2)
    m2, tok2 = loaded["model_wrapper"].model, loaded["tokenizer"]
    assert tok2.get_vocab() == dl.get_tokenizer().get_vocab()
    assert loaded["model_wrapper"].get_meta_config()["trained_on_dataset"] == "ml_1m"
    b0 = testb.batches[0]
    assert torch.equal(model(b0)["mlm_logits"].cpu(), m2(b0)["mlm_logits"].cpu())
    # apps
    history = dl.create_item_list()[:15]
    rec = Recommender(model, dl)(history)
    assert isinstance(rec, str) and rec not in history
    ranked = Ranker(model, dl)(history, dl.create_item_list()[20:30])
    assert sorted(ranked) == sorted(dl.create_item_list()[20:30])


def test_outputs_of_two_calls_do_not_alias():
    """the reference returns fresh tensors per call (bert4rec_model.py:139-149); the engine reuses one workspace per batch shape"""
    model = make_model(120, seed=4)
    b1 = orc.synthetic_batch(4, 24, 6, 120, seed=1)
    b2 = orc.synthetic_batch(4, 24, 6, 120, seed=2)
    o1 = model(b1)
    keep = {k: (v.clone() if torch.is_tensor(v) else [t.clone() for t in v]) for k, v in o1.items()}
    o2 = model(b2)
    for k in ("sequence_output", "pooled_output", "mlm_logits"):
        assert torch.equal(o1[k], keep[k]) and not torch.equal(o1[k], o2[k]), k
    assert all(torch.equal(a, b) for a, b in zip(o1["encoder_outputs"], keep["encoder_outputs"]))
    e1 = model.encoder({"input_word_ids": b1["input_word_ids"], "input_mask": b1["input_mask"]})
    e2 = model.encoder({"input_word_ids": b2["input_word_ids"], "input_mask": b2["input_mask"]})
    assert torch.equal(e1["sequence_output"], keep["sequence_output"]) and not torch.equal(e1["sequence_output"], e2["sequence_output"])


def test_rank_items_orders_like_the_reference():
    """bert4rec_model.py:203-240: gather candidate logits, stable descending argsort, gather candidates."""
    model = make_model(300, seed=9)
    batch = orc.synthetic_batch(6, 24, 6, 300, seed=2, ragged=True, finetune=True)
    rng = np.random.default_rng(0)
    items = [[rng.permutation(np.arange(3, 300))[:101].tolist()] for _ in range(6)]
    rankings = model.rank_items(batch, items)
    assert len(rankings) == 6 and all(len(r) == 1 and r[0].shape == (101,) for r in rankings)   # bert4rec_model_tests.py:127-139
    cfg_o, params = oracle_of(model)
    logits = orc.model_forward(params, batch, cfg_o)["mlm_logits"]
    # the ranked slots' transform output: only the R = 6 valid slots are transformed (no [B*P, V] logits anywhere) ...
    hid_t, slots, counts = model._ranked_slot_hidden(batch)
    assert hid_t.shape == (6, 64) and counts == [1] * 6 and slots.cpu().tolist() == [6 * b for b in range(6)]
    hidden = hid_t.cpu().numpy()
    # ... and equals what the full forward (all B*P slots) holds for those slots
    model(batch)
    full_hidden = model.engine.region("mlm_hidden", 6, 24, 6).view(6, 6, 64)[:, 0].cpu().numpy()
    assert np.abs(full_hidden - hidden).max() < 1e-6
    E = params["word_embeddings/embeddings"].numpy()
    bias = params["cls/predictions/output_bias/bias"].numpy()
    for b in range(6):
        cand = np.array(items[b], dtype=np.int64)
        got = rankings[b][0].cpu().numpy()
        # (1) bit-exact at the kernel boundary: same hidden -> identical ordering as the oracle's fma-chain scores
        want, _ = orc.rank_candidates(orc.candidate_scores_fma(hidden[b:b + 1], E, bias, cand), cand)
        assert np.array_equal(got, want[0])
        # (2) end to end against the oracle's own logits: a permutation whose scores never increase by more than 2e-4
        s = logits[b, 0, torch.from_numpy(got)].numpy()
        assert sorted(got.tolist()) == sorted(cand[0].tolist()) and (np.diff(s) < 2e-4).all()
    full = model.rank_items(batch)                                          # items=None: whole vocabulary
    assert full[0][0].shape == (300,) and sorted(full[0][0].cpu().tolist()) == list(range(300))
    s = logits[0, 0, full[0][0].cpu()].numpy()
    assert (np.diff(s) < 2e-4).all()
    ragged = model.rank_items(batch, [[[5, 6, 7]], [[8, 9]], [[10]], [[11, 12, 13, 14]], [[3, 4]], [[20, 21, 22]]])
    assert [r[0].numel() for r in ragged] == [3, 2, 1, 4, 2, 3]


def test_evaluator_ranks_equal_oracle_ranks_for_given_candidates():
    model = make_model(300, seed=11)
    batch = orc.synthetic_batch(16, 24, 6, 300, seed=5, ragged=True, finetune=True)
    rng = np.random.default_rng(1)
    gt = batch["masked_lm_ids"][:, 0].numpy()
    cands = []
    for b in range(16):
        seen = set(batch["labels"][b].tolist()) | {int(gt[b])}
        pool = [i for i in range(3, 300) if i not in seen]
        cands.append(rng.permutation(pool)[:100].tolist() + [int(gt[b])])
    cands = np.array(cands, dtype=np.int64)
    ev = evaluation.get(sampler=dataloaders.samplers.get("random", vocab=list(range(300)), sample_size=100))
    ranks = ev.evaluate_batch(model, batch, candidates=cands, ground_truth=gt)
    cfg_o, params = oracle_of(model)
    logits = orc.model_forward(params, batch, cfg_o)["mlm_logits"][:, 0].numpy()
    sc = np.take_along_axis(logits, cands, axis=1)
    ranking, _ = orc.rank_candidates(sc, cands)
    want = orc.rank_of_ground_truth(ranking, gt)
    margin = np.abs(sc - sc[:, -1:])[:, :-1].min(axis=1)          # closest competitor of the ground truth
    safe = margin > 1e-4
    assert safe.sum() >= 12 and np.array_equal(torch.as_tensor(ranks).cpu().numpy()[safe], want[safe])
    assert ev.get_metrics_results()["Valid Ranks"] == 16
    om = orc.EvalMetrics()
    for r in ranks.tolist():
        om.update(int(r))
    for k, v in om.results().items():
        assert ev.get_metrics_results()[k] == pytest.approx(v, abs=1e-12)


def test_evaluator_device_sampling_matches_host_sampling_in_distribution():
    """one b4r_sample_candidates launch per batch instead of one np.random.choice per slot.  Same law,
    other random stream: candidate lists obey the same constraints and the metrics agree statistically."""
    V, B, L = 400, 64, 20
    model = make_model(V, seed=13)
    rng = np.random.default_rng(3)
    pop = rng.zipf(1.3, size=20000) % (V - 3) + 3                       # skewed item popularity over real items
    smp = dataloaders.samplers.get("pop_random", source=pop.tolist(), vocab=list(range(V)), sample_size=100)
    batches = [orc.synthetic_batch(B, L, 4, V, seed=50 + i, ragged=True, finetune=True) for i in range(6)]
    ev_dev = evaluation.get(sampler=smp, device_sampling=True, seed=1)
    assert ev_dev._device_sampler_ready(model)
    # a sampler seeded the reference's way keeps its own numpy stream (equal seeds, equal samples): no device sampling for it
    seeded = dataloaders.samplers.get("pop_random", source=pop.tolist(), vocab=list(range(V)), sample_size=100, seed=7)
    assert not evaluation.get(sampler=seeded, device_sampling=True)._device_sampler_ready(model)
    cand, gt = ev_dev.sample_candidates_device(model, batches[0])
    cand, gt = cand.cpu().numpy(), gt.cpu().numpy()
    assert cand.shape == (B, 101) and np.array_equal(cand[:, 100], gt)
    p = np.asarray(smp.probability_distribution)
    for b in range(B):
        neg = cand[b, :100]
        assert len(set(neg.tolist())) == 100 and (p[neg] > 0).all()
        assert not (set(neg.tolist()) & (set(batches[0]["labels"][b].tolist()) | {int(gt[b])}))
    # popular items are drawn more often: mean popularity of the negatives far above the uniform mean
    assert p[cand[:, :100]].mean() > 2 * p[p > 0].mean()
    ev_host = evaluation.get(sampler=smp, device_sampling=False)
    for bt in batches:
        ev_dev.evaluate_batch(model, bt)
        ev_host.evaluate_batch(model, bt)
    rd, rh = ev_dev.get_metrics_results(), ev_host.get_metrics_results()
    assert rd["Valid Ranks"] == rh["Valid Ranks"] == 6 * B
    # an untrained model ranks the ground truth anywhere among 101: both estimates of HR@10 sit near 10/101, within noise
    for k in ("HR@10", "NDCG@10", "MAP"):
        assert abs(rd[k] - rh[k]) < 0.06, (k, rd[k], rh[k])


def test_evaluator_reports_rows_without_enough_drawable_items_and_reads_nothing_out_of_range():
    """popular_random_sampler.py:104-109 raises ValueError when the exclusions leave fewer than sample_size items.  The device
    sampler marks such rows with -1 candidates: the ranking kernel must score them -inf WITHOUT reading table[-1] / bias[-1]
    (the item table is the first tensor of the parameter buffer), and the evaluator raises once, with nothing left behind."""
    V, B, L = 112, 8, 20     # 109 real items, ~20 of them excluded per user: fewer than the 100 negatives asked for
    model = make_model(V, seed=5)
    rng = np.random.default_rng(2)
    pop = rng.integers(3, V, size=4000)
    smp = dataloaders.samplers.get("pop_random", source=pop.tolist(), vocab=list(range(V)), sample_size=100)
    ev = evaluation.get(sampler=smp, device_sampling=True, seed=3)
    assert ev._device_sampler_ready(model)
    batch = orc.synthetic_batch(B, L, 4, V, seed=9, ragged=True, finetune=True)
    ev.evaluate_batch(model, batch)            # enqueues sampling + ranking of rows full of -1 candidates
    torch.cuda.synchronize()                   # a fault would surface here
    with pytest.raises(ValueError):
        ev.get_metrics_results()
    assert ev.get_metrics_results()["Valid Ranks"] == 0      # the aborted evaluation left nothing in the accumulators
    # ids beyond the vocabulary given to rank_items score -inf: they rank behind every real item
    ranked = model.rank_items(batch, [[[5, V + 7, 6, -3]]] * B)
    for r in ranked:
        assert sorted(r[0][:2].tolist()) == [5, 6] and sorted(r[0][2:].tolist()) == [-3, V + 7]


def test_ndcg_after_training_matches_the_oracle_trained_the_same_way():
    """BASELINE.json north_star: NDCG@10 within +-0.002 of the reference.  The oracle stands in for the TF2 path:
    both sides start from the same weights, take the same AdamW steps on the same masked batches of a Zipf-popularity log
    (dropout off -- TF's dropout stream is not reproducible) and are evaluated leave-one-out against the same 100 negatives
    per user (bert4rec_evaluator.py:60-120)."""
    from bert4rec_amd.engine import make_adamw_config
    dl = make_loader()
    train, val, test = dl.prepare_training()
    V = dl.tokenizer.get_vocab_size()
    model = make_model(V, dropout=0.0)
    cfg_o, params = oracle_of(model)
    hp_o = orc.AdamWConfig(init_lr=2e-3, num_warmup_steps=5, num_train_steps=400)
    hp = make_adamw_config(hp_o.init_lr, hp_o.num_train_steps, hp_o.num_warmup_steps, hp_o.end_lr, hp_o.weight_decay_rate,
                           hp_o.beta_1, hp_o.beta_2, hp_o.epsilon, hp_o.gradient_clip_norm)
    tb = dataloaders.make_batches(train, batch_size=64, seed=1)
    m, v = orc.zeros_like_params(params), orc.zeros_like_params(params)
    eng = model.engine
    eng.set_step(0)
    step = 0
    for _ in range(4):                      # epochs over the cached batches, like trainer.train
        for b in tb.batches:
            b = {k: torch.as_tensor(t).cpu() for k, t in b.items()}
            orc.train_step(params, m, v, b, cfg_o, hp_o, step=step, training=False)
            cb, keep = eng.prepare_batch(b)
            eng.train_step(hp, cb)
            step += 1
    torch.cuda.synchronize()
    got_w = model.get_weights()
    drift = max(float((got_w[n].cpu() - p.reshape(got_w[n].shape)).abs().max()) for n, p in params.items() if orc.is_trainable(n))
    assert drift < 1e-4, drift              # the two trajectories stay together over all steps

    ev = evaluation.get(sampler=dataloaders.samplers.get("random", vocab=list(range(V)), sample_size=100))
    om = orc.EvalMetrics()
    rng = np.random.default_rng(7)
    n_users = 0
    for b in dataloaders.make_batches(test, batch_size=64, seed=1).batches:
        b = {k: torch.as_tensor(t).cpu() for k, t in b.items()}
        gt = b["masked_lm_ids"][:, 0].numpy()
        cands = []
        for r in range(len(gt)):
            seen = set(b["labels"][r].tolist()) | {int(gt[r])}
            pool = [i for i in range(3, V) if i not in seen]
            cands.append(rng.permutation(pool)[:100].tolist() + [int(gt[r])])
        cands = np.array(cands, dtype=np.int64)
        ev.evaluate_batch(model, b, candidates=cands, ground_truth=gt)
        logits = orc.model_forward(params, b, cfg_o)["mlm_logits"][:, 0].numpy()
        ranking, _ = orc.rank_candidates(np.take_along_axis(logits, cands, axis=1), cands)
        for rnk in orc.rank_of_ground_truth(ranking, gt).tolist():
            om.update(int(rnk))
        n_users += len(gt)
    got, want = ev.get_metrics_results(), om.results()
    assert got["Valid Ranks"] == n_users == len(test)
    for k in ("NDCG@10", "HR@10", "NDCG@5", "MAP"):
        assert abs(got[k] - want[k]) <= 0.002, (k, got[k], want[k])
    assert want["NDCG@10"] > 0.07           # the 4 epochs learned something: chance level is 0.045 with 100 negatives


def test_device_masked_batches_follow_the_reference_contract():
    """SURVEY.md §8 f1 wired in: prepare_training(device_masking=True) + make_batches -> batches masked by b4r_mask_batch.
    Deterministic parts are compared bit for bit with goldens captured from the reference (finetuning branch = last-token mask,
    padding, counts); the random part by its law (count formula, ascending positions, labels)."""
    import json
    import os
    from bert4rec_amd.engine import device_mask_batch
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_goldens.json")))
    g = gold["process_element_no_mlm"]
    L, P = g["max_seq_len"], g["max_predictions_per_seq"]
    rows = torch.tensor([c["labels"] for c in g["cases"]], dtype=torch.int64)
    out = {k: v.cpu() for k, v in device_mask_batch(rows.cuda(), P, len(g["vocab"]), finetune=True, seed=5).items()}
    for r, c in enumerate(g["cases"]):
        n = sum(c["input_mask"])
        want_ids = list(c["labels"]); want_ids[n - 1] = 1                     # mask_last_token_only + right padding
        assert out["input_word_ids"][r].tolist() == want_ids and out["labels"][r].tolist() == c["labels"]
        assert out["input_mask"][r].tolist() == c["input_mask"]
        assert out["masked_lm_positions"][r].tolist() == [n - 1] + [0] * (P - 1)
        assert out["masked_lm_ids"][r].tolist() == [c["labels"][n - 1]] + [0] * (P - 1)
        assert out["masked_lm_weights"][r].tolist() == [1] + [0] * (P - 1)
    for c in gold["mask_last_token_only"]:
        t = torch.tensor([c["sequence"] + [0] * (L - len(c["sequence"]))], dtype=torch.int64)
        o = device_mask_batch(t.cuda(), P, 100, finetune=True)
        assert o["input_word_ids"][0, :len(c["sequence"])].cpu().tolist() == c["masked_token_ids"]
        assert o["masked_lm_positions"][0, 0].item() == c["masked_lm_positions"][0]

    # end to end through the dataloader: 90 / 10 split by per-row flags, frozen masks by default, new masks with remask
    ds = datasets.synthetic_dataset(n_users=60, n_items=300, min_len=2, max_len=40, seed=1)
    dl = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=ds, max_seq_len=20,
                                                                               max_predictions_per_seq=5, input_duplication_factor=2)
    train, val, test = dl.prepare_training(finetuning_split=0.1, device_masking=True)
    frozen = dataloaders.make_batches(train, batch_size=32, seed=3).cache_on_device("cuda")
    e1 = [{k: v.cpu() for k, v in b.items()} for b in frozen]
    e2 = [{k: v.cpu() for k, v in b.items()} for b in frozen]
    assert len(e1) == 4 and [b["input_word_ids"].shape[0] for b in e1] == [32, 32, 32, 24]
    assert all(torch.equal(a[k], b[k]) for a, b in zip(e1, e2) for k in a)        # == tf .cache(): masks frozen after epoch 1
    assert set(e1[0]) == {"labels", "input_word_ids", "input_mask", "masked_lm_ids", "masked_lm_positions", "masked_lm_weights"}
    n_ft = 0
    for b in e1:
        for r in range(b["labels"].shape[0]):
            n = int(b["input_mask"][r].sum())
            plain = int(((b["labels"][r] != 0) & (b["labels"][r] != 2)).sum())
            k = int(b["masked_lm_weights"][r].sum())
            pos = b["masked_lm_positions"][r, :k]
            last_only = k == 1 and int(pos[0]) == n - 1 and plain * 0.2 >= 2    # a dynamic row would have drawn >= 2 positions
            n_ft += int(last_only)
            assert k == 1 or k == min(5, max(1, int(plain * 0.2)))
            assert pos.tolist() == sorted(pos.tolist()) and (b["input_word_ids"][r, pos] == 1).all()
            assert torch.equal(b["labels"][r, pos], b["masked_lm_ids"][r, :k])
            assert (b["masked_lm_positions"][r, k:] == 0).all() and (b["masked_lm_ids"][r, k:] == 0).all()
    assert 1 <= n_ft <= int(train.finetune_rows.sum())
    remask = dataloaders.make_batches(train, batch_size=32, seed=3, remask_each_epoch=True).cache_on_device("cuda")
    r1 = [b["masked_lm_positions"].cpu() for b in remask]
    r2 = [b["masked_lm_positions"].cpu() for b in remask]
    assert any(not torch.equal(a, b) for a, b in zip(r1, r2))                     # new masks ...
    assert all(torch.equal(a["labels"], b["labels"].cpu()) for a, b in zip(e1, remask))   # ... same batch composition
    # validation / test rows: last-token mask only; the model consumes device-masked batches directly
    vb = next(iter(dataloaders.make_batches(val, batch_size=16, seed=1).cache_on_device("cuda")))
    assert (vb["masked_lm_weights"].sum(1) == 1).all()
    model = make_model(dl.tokenizer.get_vocab_size(), seed=2, L=20)
    model.compile()
    hist = model.fit(remask, validation_data=dataloaders.make_batches(val, batch_size=16, seed=1), epochs=2, verbose=0)
    assert len(hist.history["loss"]) == 2 and all(np.isfinite(hist.history["loss"]))

    # bucketed + trimmed batches (not in the reference): what is cut is padding only, and the model trains on the mixed widths
    ds2 = datasets.synthetic_dataset(n_users=96, n_items=300, min_len=2, max_len=70, seed=4)
    dl2 = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=ds2, max_seq_len=64,
                                                                                max_predictions_per_seq=12, input_duplication_factor=1)
    tr2, va2, _ = dl2.prepare_training(finetuning_split=0.1, device_masking=True)
    wide = dataloaders.make_batches(tr2, batch_size=16, seed=5, bucket_by_length=3).cache_on_device("cuda")
    cut = dataloaders.make_batches(tr2, batch_size=16, seed=5, bucket_by_length=3, trim_padding=True).cache_on_device("cuda")
    widths = set()
    for a, b in zip(wide, cut):
        w, p = b["input_word_ids"].shape[1], b["masked_lm_ids"].shape[1]
        widths.add((w, p))
        assert w % 16 == 0 and p % 4 == 0 and (w < 64 or p < 12 or (w, p) == (64, 12))
        for k in a:
            n = w if a[k].shape[1] == 64 else p
            assert torch.equal(a[k][:, :n], b[k]) and int(a[k][:, n:].abs().sum()) == 0, k   # same masks, only zeros are cut
    assert len(widths) > 1
    model2 = make_model(dl2.tokenizer.get_vocab_size(), seed=2, L=64)
    model2.compile()
    h2 = model2.fit(cut, validation_data=dataloaders.make_batches(va2, batch_size=16, seed=1, trim_padding=True), epochs=2, verbose=0)
    assert len(h2.history["loss"]) == 2 and all(np.isfinite(h2.history["loss"]))


def test_apps_return_what_the_oracle_ranks_first():
    """apps/recommender.py:14-63 and apps/ranker.py:19-76 on values: the recommended item is the oracle's arg-max over the unseen
    vocabulary for the masked slot, the Ranker's order is the oracle's descending-logit order (best first: the reference's negated
    logits, ranker.py:29, are not reproduced)."""
    from bert4rec_amd.apps import Ranker, Recommender
    ds = datasets.synthetic_dataset(n_users=30, n_items=200, min_len=5, max_len=30, seed=4)
    dl = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=ds, max_seq_len=24, max_predictions_per_seq=6)
    dl.generate_vocab()
    V = dl.tokenizer.get_vocab_size()
    model = make_model(V, seed=5)
    cfg_o, params = oracle_of(model)
    items_all = dl.create_item_list()
    for start in (0, 40, 90):
        history = items_all[start:start + 15]
        batch = {k: torch.from_numpy(np.asarray(v)) for k, v in dl.prepare_inference(list(history)).items()}
        logits = orc.model_forward(params, batch, cfg_o)["mlm_logits"][0, 0].numpy()     # slot 0 = the appended masked token
        seen = set(dl.tokenizer.tokenize(list(history))) | {0, 1, 2}
        order = [i for i in np.argsort(-logits.astype(np.float64), kind="stable") if i not in seen]
        margin = logits[order[0]] - logits[order[1]]
        rec = Recommender(model, dl)(history)
        assert rec not in history
        if margin > 1e-4:
            assert rec == dl.tokenizer.detokenize(int(order[0]))
        top5 = Recommender(model, dl)(history, k=5)
        gaps = np.diff(-logits[order[:6]])
        if (gaps > 1e-4).all():
            assert top5 == dl.tokenizer.detokenize([int(i) for i in order[:5]])
        cands = [it for it in dict.fromkeys(items_all[start + 20:start + 60]) if it not in history][:10]
        ranked = Ranker(model, dl)(history, cands)
        ids = dl.tokenizer.tokenize(list(cands))
        sc = logits[ids]
        want = [cands[j] for j in np.argsort(-sc.astype(np.float64), kind="stable")]
        assert sorted(ranked) == sorted(cands)
        if (np.abs(np.diff(np.sort(sc))) > 1e-4).all():
            assert ranked == want


def test_custom_weight_decay_selection_decays_exactly_the_selected_variables():
    """AdamWeightDecay._do_use_weight_decay (adam_w_optimizer.py:154-168) with a non-default exclusion list: the flat buffer's built-in
    rule no longer applies, the optimizer kernel takes a per-element mask.  One step from the same weights on the same batch with
    exclude=["bias"] and with the default list: the LayerNorm variables (now decayed) differ by -lr * rate * value, nothing else."""
    from bert4rec_amd.trainers import optimizers
    batch = orc.synthetic_batch(8, 24, 5, 60, seed=1)
    out = []
    for excl in (None, ["bias"]):
        model = make_model(60, seed=2)
        opt = optimizers.get("adamw", init_lr=1e-2, num_warmup_steps=0, weight_decay_rate=0.1, exclude_from_weight_decay=excl)
        model.compile(optimizer=opt)
        before = {k: v.clone() for k, v in model.get_weights().items()}
        model.train_step(batch)
        out.append((before, model.get_weights(), model._hp))
    (b0, w0, hp0), (b1, w1, hp1) = out
    assert not hp0.decay_mask and hp1.decay_mask
    lr = 1e-2   # step 0 of PolynomialDecay without warm-up
    n_ln = 0
    for k in w0:
        assert torch.equal(b0[k], b1[k])
        d = (w1[k] - w0[k]).double()
        if "layer_norm" in k or "LayerNorm" in k:
            n_ln += 1
            want = (-lr * 0.1 * b0[k]).double()
            assert float((d - want).abs().max()) < 1e-6 * max(1.0, float(b0[k].abs().max())), k
        else:
            assert float(d.abs().max()) < 1e-7, k     # (the item table's gradient is summed with float atomics: last-bit noise)
    assert n_ln >= 6


def test_evaluation_on_trimmed_batches_gives_the_same_ranks():
    """make_batches(trim_padding=True) on the test split: the evaluator's candidate draws depend on (seed, draw counter, row), the
    ranking on the hidden state of each user's last position -- neither on the padding columns that are cut, so the ground-truth
    ranks and every metric are those of the padded batches."""
    ds = datasets.synthetic_dataset(n_users=96, n_items=300, min_len=4, max_len=70, seed=4)
    dl = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=ds, max_seq_len=64,
                                                                               max_predictions_per_seq=12, input_duplication_factor=1)
    _, _, test = dl.prepare_training(finetuning_split=0.1)
    model = make_model(dl.tokenizer.get_vocab_size(), seed=2, L=64)
    smp = dataloaders.samplers.get("pop_random", source=[t for e in test.examples for t in e["labels"].tolist() if t > 2],
                                   vocab=list(range(dl.tokenizer.get_vocab_size())), sample_size=50, seed=3)
    results, ranks = [], []
    for trim in (False, True):
        ev = evaluation.get(sampler=smp, device_sampling=True, seed=5)
        bs = dataloaders.make_batches(test, batch_size=16, seed=1, bucket_by_length=6, trim_padding=trim)   # same composition both times
        if trim:
            assert min(b["input_word_ids"].shape[1] for b in bs) < 64 and all(b["masked_lm_ids"].shape[1] == 4 for b in bs)
        ranks.append(torch.cat([ev.evaluate_batch(model, b).cpu() for b in bs]))
        results.append(ev.get_metrics_results())
    same = (ranks[0] == ranks[1]).float().mean().item()
    assert same > 0.99, same                     # (a rank can move by one where two candidate scores agree to ~1e-6)
    for k in results[0]:
        assert abs(results[0][k] - results[1][k]) < 0.01 * max(1.0, abs(results[0][k])), k


def test_evaluation_of_resident_batches_equals_evaluation_of_host_batches():
    """BatchedDataset.cache_on_device keeps the batches in HBM together with the (row, slot) pairs that carry a weight (found on the
    host copy).  The evaluator then needs no read-back per batch, keeps each batch's constants (slots, ground truth, exclusion
    rows, ranked rows) from the first pass on, and lets the model run the last layer's feed-forward half on the ranked rows only
    (B4R_FLAG_ENCODER_ONLY | B4R_FLAG_HEAD_ROWS_ONLY).  Same draws (seed, draw counter, row), same ranks as with host batches --
    on the first pass (constants formed) and on the second (constants reused)."""
    ds = datasets.synthetic_dataset(n_users=160, n_items=300, min_len=4, max_len=60, seed=6)
    dl = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=ds, max_seq_len=64,
                                                                               max_predictions_per_seq=12, input_duplication_factor=1)
    _, _, test = dl.prepare_training(finetuning_split=0.1)
    V = dl.tokenizer.get_vocab_size()
    model = make_model(V, seed=4, L=64)
    source = [t for e in test.examples for t in e["labels"].tolist() if t > 2]
    smp = dataloaders.samplers.get("pop_random", source=source, vocab=list(range(V)), sample_size=50)
    host = dataloaders.make_batches(test, batch_size=32, seed=1)
    resident = dataloaders.make_batches(test, batch_size=32, seed=1).cache_on_device("cuda")
    first = next(iter(resident))
    assert first["input_word_ids"].is_cuda and first.slot_index.shape[1] == 2 and set(first) == set(next(iter(host)))
    runs = []
    for bs, passes in ((host, 1), (resident, 2)):
        for _ in range(passes):
            ev = evaluation.get(sampler=smp, device_sampling=True, seed=5)
            ranks = torch.cat([ev.evaluate_batch(model, b).cpu() for b in bs])
            runs.append((ranks, ev.get_metrics_results()))
    assert first.eval_cache is not None and first.eval_cache[3] is not None      # ... and the rows-only forward was allowed
    for ranks, res in runs[1:]:
        assert res["Valid Ranks"] == runs[0][1]["Valid Ranks"] == 160
        same = (ranks == runs[0][0]).float().mean().item()
        assert same > 0.99, same                 # (a rank can move by one where two candidate scores agree to ~1e-6)
    assert torch.equal(runs[1][0], runs[2][0])   # constants formed vs constants reused: bit for bit
    # a row that cannot supply its negatives is reported through the flag the sampler kernel sets itself
    few = dataloaders.samplers.get("pop_random", source=source[:40], vocab=list(range(V)), sample_size=290)
    ev = evaluation.get(sampler=few, device_sampling=True, seed=5)
    if ev._device_sampler_ready(model):
        for b in resident:
            ev.evaluate_batch(model, b)
        with pytest.raises(ValueError):
            ev.get_metrics_results()


def test_ml_1m_example_runs_with_the_reference_literals_for_one_epoch(tmp_path, monkeypatch):
    """examples/bert4rec_ml_1m_example.py: defaults = the reference's literals (150 epochs, ml-1m_128.json, duplication 5, patience
    20: /root/reference/examples/bert4rec_ml_1m_example.py:21-30); here one epoch of that exact configuration on the synthetic
    fallback log -- the H = 128 / 4-head / inner-512 encoder the reference's own ML-1M run trains."""
    import importlib.util
    import inspect
    import pathlib
    path = pathlib.Path(__file__).resolve().parent.parent / "examples" / "bert4rec_ml_1m_example.py"
    spec = importlib.util.spec_from_file_location("b4r_ml1m_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = {k: v.default for k, v in inspect.signature(mod.main).parameters.items()}
    assert (d["epochs"], d["batch_size"], d["input_duplication_factor"], d["finetuning_split"], d["encoder_config"], d["patience"],
            d["append_early_stopping"]) == (150, 256, 5, 0.1, "ml-1m_128.json", 20, False)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("B4R_DATA_DIR", str(tmp_path / "no_data_here"))
    metrics = mod.main(epochs=1, input_duplication_factor=1)
    assert metrics["Valid Ranks"] == 2000 and 0 <= metrics["NDCG@10"] <= 1
    out = tmp_path / "bert4rec_ml-1m_15"
    assert (out / "model_weights.safetensors").is_file() and (out / "eval_results.json").is_file() and (out / "vocab.txt").is_file()
    import json
    meta = json.load(open(out / "meta_config.json"))
    assert meta["EPOCHS"] == 1 and meta["encoder_config"]["hidden_size"] == 128 and meta["early_stopping_config"]["patience"] == 20
