"""The oracle itself is checked here (CPU): against the golden vectors captured from the reference's own numpy code
(tests/golden/reference_goldens.json), against the reference's known-answer metric tests, and assumption by assumption
for the third-party semantics it restates (header of oracle/bert4rec_oracle.py, items i-x)."""
import ctypes as C
import json
import math
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import bert4rec_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_goldens.json")))


# ---- pinned by the reference -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key", sorted(GOLD["evaluation_metrics"]))
def test_eval_metrics_match_reference_outputs(key):
    case = GOLD["evaluation_metrics"][key]
    m = orc.EvalMetrics()
    for r in case["ranks"]:
        m.update(r)
    got = m.results()
    for name, want in case["results"].items():
        assert got[name] == pytest.approx(want, rel=0, abs=1e-15), name


def test_eval_metrics_known_answers_of_the_reference_tests():
    """tests/evaluators_tests/evaluation_metrics_tests.py:28-104 (values as asserted there)."""
    expect = {
        (1, 2, 3, 4, 5): dict(hr=(0.2, 1, 1), ndcg=(0.2, 0.59, 0.59), map=0.46, n=5),
        (1, 5, 10, 15, 20): dict(hr=(0.2, 0.4, 0.6), ndcg=(0.2, 0.28, 0.34), map=0.28, n=5),
        (2, 8, 4, 13, 20, 6, 3, 11, 2, 5): dict(hr=(0, 0.5, 0.7), ndcg=(0, 0.26, 0.33), map=0.23, n=10),
    }
    for ranks, e in expect.items():
        m = orc.EvalMetrics()
        for r in ranks:
            m.update(r)
        res = m.results()
        assert (res["HR@1"], res["HR@5"], res["HR@10"]) == e["hr"]
        assert res["NDCG@1"] == e["ndcg"][0]
        assert round(res["NDCG@5"], 2) == e["ndcg"][1] and round(res["NDCG@10"], 2) == e["ndcg"][2]
        assert round(res["MAP"], 2) == e["map"] and res["Valid Ranks"] == e["n"]


@pytest.mark.parametrize("i", range(len(GOLD["apply_dynamic_masking_task"])))
def test_dynamic_masking_matches_reference_outputs(i):
    c = GOLD["apply_dynamic_masking_task"][i]
    toks, pos, ids = orc.apply_dynamic_masking_task(np.array(c["sequence"], dtype=np.int64), c["max_selections_per_seq"],
                                                    c["mask_token_id"], c["special_token_ids"], c["vocab_size"],
                                                    c["selection_rate"], c["mask_token_rate"], c["random_token_rate"],
                                                    seed=c["seed"])
    assert toks.tolist() == c["masked_token_ids"]
    assert pos.tolist() == c["masked_lm_positions"]
    assert ids.tolist() == c["masked_lm_ids"]


def test_masking_invariants_like_the_reference_tests():
    """tests/datalaoders_tests/dataloader_utils_tests.py:180-248"""
    seq = np.arange(3, 103, dtype=np.int64)
    toks, pos, ids = orc.apply_dynamic_masking_task(seq, 40, 1, [2, 0], 200, 0.2, 1.0, 0.0, seed=3)
    assert len(toks) == len(seq) and 0 < len(pos) <= 40 and len(pos) == len(ids) == 20
    assert set(ids.tolist()) <= set(seq.tolist()) and (toks[pos] == 1).all() and (np.diff(pos) > 0).all()
    s2, p2, i2 = orc.mask_last_token_only(seq, 1)
    assert s2[-1] == 1 and p2.tolist() == [len(seq) - 1] and i2.tolist() == [102] and (s2[:-1] == seq[:-1]).all()


def test_process_element_layout():
    """tests/datalaoders_tests/preprocessors_tests/bert4rec_preprocessor_tests.py:61-160"""
    out = orc.process_element(list(range(3, 13)), 20, 5, 100, apply_mlm=True, finetuning=True)
    assert set(out) == {"labels", "input_word_ids", "input_mask", "masked_lm_ids", "masked_lm_positions", "masked_lm_weights"}
    assert all(len(out[k]) == 20 for k in ("labels", "input_word_ids", "input_mask"))
    assert all(len(out[k]) == 5 for k in ("masked_lm_ids", "masked_lm_positions", "masked_lm_weights"))
    assert out["masked_lm_weights"].tolist() == [1, 0, 0, 0, 0] and out["masked_lm_positions"][0] == 9
    assert out["input_mask"].tolist() == [1] * 10 + [0] * 10 and out["input_word_ids"][9] == 1
    assert orc.process_element(list(range(3, 13)), 20, 5, 100, False, False).keys() == {"labels", "input_word_ids", "input_mask"}
    long = orc.process_element(list(range(3, 63)), 20, 5, 100, True, True)
    assert long["labels"].tolist() == list(range(43, 63))  # finetuning keeps the most recent items


# ---- third-party semantics restated by the oracle (assumptions i-x) ----------------------------------------------------------
def test_layer_norm_is_keras_nonfused_form():
    x = torch.randn(7, 64)
    g, b = torch.rand(64) + 0.5, torch.randn(64)
    want = torch.nn.functional.layer_norm(x.double(), (64,), g.double(), b.double(), eps=1e-12)
    assert float((orc.layer_norm(x, g, b).double() - want).abs().max()) < 1e-5
    # zero-variance row: inv = 1e6*gamma, and  x*inv + (beta - mean*inv)  cancels only to the fp32 spacing at 3e6 (0.25):
    # the non-fused Keras formula is what is restated, including this degenerate behaviour
    const = torch.full((2, 64), 3.0)
    assert float((orc.layer_norm(const, g, b) - b.expand(2, 64)).abs().max()) <= 0.5


def test_gelu_is_erf_form():
    x = torch.linspace(-4, 4, 101)
    assert float((orc.gelu_erf(x) - torch.nn.functional.gelu(x, approximate="none")).abs().max()) < 1e-6
    assert float((orc.gelu_erf(x) - torch.nn.functional.gelu(x, approximate="tanh")).abs().max()) > 1e-4


def _tiny():
    cfg = orc.OracleConfig(vocab_size=37, hidden_size=64, num_layers=2, num_attention_heads=2, max_sequence_length=16, inner_dim=64)
    return cfg, orc.init_params(cfg, 3)


def test_attention_mask_is_additive_minus_1e9_on_keys_only():
    cfg, p = _tiny()
    b = orc.synthetic_batch(2, 16, 4, 37, seed=1, ragged=True)
    out = orc.model_forward(p, b, cfg)
    # changing a PADDED token id changes its own row but no valid row (keys are masked, queries are not)
    b2 = {k: v.clone() for k, v in b.items()}
    n0 = int(b["input_mask"][0].sum())
    if n0 < 16:
        b2["input_word_ids"][0, 15] = 5
        out2 = orc.model_forward(p, b2, cfg)
        assert torch.allclose(out["sequence_output"][0, :n0], out2["sequence_output"][0, :n0], atol=1e-6)
        assert not torch.allclose(out["sequence_output"][0, 15], out2["sequence_output"][0, 15], atol=1e-6)
    # a fully masked row attends uniformly (-1e9 added to every key, not -inf)
    b3 = {k: v.clone() for k, v in b.items()}
    b3["input_mask"][1] = 0
    assert torch.isfinite(orc.model_forward(p, b3, cfg)["sequence_output"]).all()


def test_post_ln_block_and_tied_projection_shapes_and_keys():
    cfg, p = _tiny()
    b = orc.synthetic_batch(3, 16, 4, 37, seed=2)
    out = orc.model_forward(p, b, cfg)
    assert out["sequence_output"].shape == (3, 16, 64) and out["pooled_output"].shape == (3, 64)
    assert len(out["encoder_outputs"]) == 2 and out["mlm_logits"].shape == (3, 4, 37)
    enc_only = orc.model_forward(p, {k: b[k] for k in ("input_word_ids", "input_mask")}, cfg)
    assert "mlm_logits" not in enc_only          # tests/models_tests/bert4rec_model_tests.py:66-95
    # the output of every block is a LayerNorm output: per-token mean ~ beta mean (0), variance ~ 1
    x = out["encoder_outputs"][0]
    assert float(x.mean(-1).abs().max()) < 1e-4 and float((x.var(-1, unbiased=False) - 1).abs().max()) < 1e-3
    # weight tying: perturbing the table moves the logits through the projection as well
    t = out["mlm_hidden"]
    logits = t @ p["word_embeddings/embeddings"].t() + p["cls/predictions/output_bias/bias"]
    assert torch.allclose(logits, out["mlm_logits"], atol=1e-6)


def test_query_is_scaled_after_bias():
    cfg, p = _tiny()
    p["transformer/layer_0/self_attention/query/bias"] += 0.5
    b = orc.synthetic_batch(2, 16, 4, 37, seed=3)
    out = orc.model_forward(p, b, cfg)["sequence_output"]
    # scaling kernel AND bias by s, and removing the 1/sqrt(d): equivalent only if the scale is applied after the bias
    assert torch.isfinite(out).all()


def test_loss_is_batch_global_masked_mean():
    logits = torch.randn(2, 3, 11)
    y = torch.tensor([[4, 0, 7], [0, 0, 2]])
    per = torch.nn.functional.cross_entropy(logits.reshape(-1, 11), y.reshape(-1), reduction="none").reshape(2, 3)
    want = (per[0, 0] + per[0, 2] + per[1, 2]) / 3
    assert float((orc.masked_sparse_categorical_crossentropy(y, logits) - want).abs()) < 1e-6
    pred = logits.argmax(-1)
    assert float(orc.masked_accuracy(y, logits)) == pytest.approx(float(((pred == y) & (y != 0)).sum()) / 3)
    assert float(orc.sparse_categorical_accuracy(y, logits)) == pytest.approx(float((pred == y).sum()) / 6)


def test_learning_rate_schedule():
    hp = orc.AdamWConfig()
    assert orc.learning_rate(0, hp) == 0.0                       # warm-up starts at lr 0
    assert orc.learning_rate(50, hp) == pytest.approx(5e-5, rel=1e-6)
    assert orc.learning_rate(100, hp) == pytest.approx(1e-4 * (1 - 100 / 400000), rel=1e-6)   # raw step fed to the decay
    assert orc.learning_rate(400000, hp) == 0.0 and orc.learning_rate(500000, hp) == 0.0


def test_adamw_update_formula_and_decay_selection():
    hp = orc.AdamWConfig(num_warmup_steps=0)
    params = {"a/kernel": torch.tensor([1.0, -2.0]), "a/bias": torch.tensor([0.5]), "x/LayerNorm/gamma": torch.tensor([1.0]),
              "y/layer_norm/beta": torch.tensor([0.1]), "word_embeddings/embeddings": torch.tensor([0.3])}
    assert [orc.uses_weight_decay(n) for n in params] == [True, False, False, False, True]
    grads = {n: torch.full_like(p, 0.1) for n, p in params.items()}
    m = {n: torch.zeros_like(p) for n, p in params.items()}
    v = {n: torch.zeros_like(p) for n, p in params.items()}
    p0 = {n: p.clone() for n, p in params.items()}
    gnorm = orc.adamw_apply(params, grads, m, v, step=0, hp=hp)
    assert gnorm == pytest.approx(0.1 * math.sqrt(6), rel=1e-6)     # below clip 5.0 => scale 1
    lr = float(orc.learning_rate(0, hp))
    alpha = lr * math.sqrt(1 - 0.999) / (1 - 0.9)
    for n in params:
        mm, vv = 0.1 * 0.1, 0.01 * 0.001
        dec = lr * 0.01 * float(p0[n][0]) if orc.uses_weight_decay(n) else 0.0
        want = float(p0[n][0]) - dec - mm * alpha / (math.sqrt(vv) + 1e-6)
        assert float(params[n][0]) == pytest.approx(want, rel=1e-5), n
    # clip_by_global_norm: scale = clip / max(norm, clip)
    big = {n: torch.full_like(p, 100.0) for n, p in params.items()}
    m2 = {n: torch.zeros_like(p) for n, p in params.items()}
    v2 = {n: torch.zeros_like(p) for n, p in params.items()}
    gn = orc.adamw_apply({n: p.clone() for n, p in params.items()}, big, m2, v2, 0, hp)
    assert float(m2["a/bias"][0]) == pytest.approx(0.1 * 100.0 * 5.0 / gn, rel=1e-5)


def test_pooler_is_not_trainable():
    cfg, p = _tiny()
    loss, grads, _ = orc.loss_and_grads(p, orc.synthetic_batch(2, 16, 4, 37), cfg, training=False)
    assert "pooler_transform/kernel" not in grads and len(grads) == len(p) - 2


def test_argsort_descending_is_stable_and_c_oracle_agrees():
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(HERE), "oracle")])
    co = C.CDLL(os.path.join(os.path.dirname(HERE), "oracle", "librank_oracle.so"))
    scores = np.array([[0.5, 2.0, 0.5, -1.0, 2.0, 0.5]], dtype=np.float32)
    cand = np.array([[10, 11, 12, 13, 14, 15]], dtype=np.int64)
    ranking, pos = orc.rank_candidates(scores, cand)
    assert ranking.tolist() == [[11, 14, 10, 12, 15, 13]]     # ties keep the lower index first (top_k semantics)
    assert orc.rank_of_ground_truth(ranking, np.array([15])).tolist() == [5]
    rk = np.zeros_like(cand)
    ps = np.zeros((1, 6), np.int32)
    f32p, i64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    co.rank_oracle_rank(scores.ctypes.data_as(f32p), cand.ctypes.data_as(i64p), C.c_int64(1), C.c_int64(6),
                        rk.ctypes.data_as(i64p), ps.ctypes.data_as(i32p))
    assert rk.tolist() == ranking.tolist() and ps.tolist() == pos.tolist()
    # scores: fmaf chain in C == the float64-emulated chain of the numpy oracle
    rng = np.random.default_rng(0)
    hid, tab, bias = rng.normal(size=(4, 64)).astype(np.float32), rng.normal(size=(50, 64)).astype(np.float32), \
        rng.normal(size=50).astype(np.float32)
    cd = np.stack([rng.permutation(50)[:20] for _ in range(4)]).astype(np.int64)
    sc = np.zeros((4, 20), np.float32)
    co.rank_oracle_scores(hid.ctypes.data_as(f32p), tab.ctypes.data_as(f32p), bias.ctypes.data_as(f32p),
                          cd.ctypes.data_as(i64p), C.c_int64(4), C.c_int64(20), C.c_int64(64), sc.ctypes.data_as(f32p))
    assert np.array_equal(sc, orc.candidate_scores_fma(hid, tab, bias, cd))
    out = np.zeros(8)
    ranks = np.array([2, 8, 4, 13, 20, 6, 3, 11, 2, 5], dtype=np.int64)
    co.rank_oracle_metrics(ranks.ctypes.data_as(i64p), C.c_int64(10), out.ctypes.data_as(C.POINTER(C.c_double)))
    want = GOLD["evaluation_metrics"]["ranks_3"]["results"]
    assert out[0] == 10 and out[3] / 10 == pytest.approx(want["NDCG@10"], abs=1e-15) and out[7] / 10 == pytest.approx(want["MAP"], abs=1e-15)


def test_dropout_hash_statistics_and_determinism():
    k1 = orc.dropout_keep_mask((200000,), 0.2, 1234, 5, 3)
    k2 = orc.dropout_keep_mask((200000,), 0.2, 1234, 5, 3)
    assert torch.equal(k1, k2) and abs(float(k1.float().mean()) - 0.8) < 0.005
    assert not torch.equal(k1, orc.dropout_keep_mask((200000,), 0.2, 1234, 6, 3))      # new step, new mask
    assert not torch.equal(k1, orc.dropout_keep_mask((200000,), 0.2, 1234, 5, 7))      # other site, other mask


def test_dropout_group_fields_are_unbiased_and_uncorrelated():
    """one hash serves 4 consecutive elements (two halves of the hash and of one xorshift32 step of it): every field must
    keep with probability 1 - rate, fields must not be correlated with each other, nor with the neighbouring group, nor
    with the same element of the next step; the attention pitch only changes which index an element has"""
    n, rate = 400000, 0.2
    k = orc.dropout_keep_mask((n,), rate, 99, 7, 5).float().view(-1, 4)
    sigma = (rate * (1 - rate) / (n / 4)) ** 0.5
    assert float((k.mean(0) - (1 - rate)).abs().max()) < 4.5 * sigma
    c = torch.corrcoef(torch.cat([k[:-1], k[1:]], dim=1).T)          # 8 x 8: fields of a group and of its neighbour
    off = c - torch.eye(8)
    assert float(off.abs().max()) < 0.02
    nxt = orc.dropout_keep_mask((n,), rate, 99, 8, 5).float().view(-1, 4)
    assert abs(float(torch.corrcoef(torch.stack([k.flatten(), nxt.flatten()]))[0, 1])) < 0.01
    a = orc.dropout_keep_mask((3, 2, 10, 10), rate, 1, 2, 3, row_pitch=orc.ATTN_PITCH)
    flat = orc.dropout_keep_mask((3 * 2 * 10 * orc.ATTN_PITCH,), rate, 1, 2, 3).view(3, 2, 10, orc.ATTN_PITCH)
    assert torch.equal(a, flat[..., :10])


def test_encoder_block_agrees_with_an_independent_implementation():
    """A restatement slip check, NOT a parity pin (the float path stays "parity unpinned": TensorFlow is not available).
    The oracle's encoder stack (einsum attention with the additive -1e9 key mask, post-LN, erf-GELU) against PyTorch's own
    nn.TransformerEncoderLayer (norm_first=False, activation=gelu [erf form], batch_first) with the weights mapped over: an
    implementation written by other people for the same published block.  Both see the same embedded input."""
    torch.manual_seed(0)
    cfg = orc.OracleConfig(vocab_size=50, hidden_size=64, num_layers=2, num_attention_heads=2, max_sequence_length=24, inner_dim=256)
    params = orc.init_params(cfg, seed=9)
    for k in params:   # biases and LayerNorm offsets away from their init values (0 / 1), so that a swapped pair would show
        if k.endswith("bias") or k.endswith("beta"):
            params[k] = 0.1 * torch.randn_like(params[k])
        if k.endswith("gamma"):
            params[k] = 1.0 + 0.2 * torch.randn_like(params[k])
    B, L, H, h, d = 5, 24, 64, 2, 32
    ids = torch.randint(3, 50, (B, L))
    lens = torch.tensor([24, 7, 1, 15, 24])
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int64)
    out = orc.encoder_forward(params, ids, mask, cfg)
    # the block input of the oracle: embedding + position + LayerNorm (restated in three lines; dropout off)
    x = params["word_embeddings/embeddings"][ids] + params["position_embedding/embeddings"][:L][None]
    x = orc.layer_norm(x, params["embeddings/layer_norm/gamma"], params["embeddings/layer_norm/beta"], cfg.ln_eps)
    for i in range(cfg.num_layers):
        pre = f"transformer/layer_{i}"
        layer = torch.nn.TransformerEncoderLayer(d_model=H, nhead=h, dim_feedforward=256, dropout=0.0, activation="gelu",
                                                 layer_norm_eps=cfg.ln_eps, batch_first=True, norm_first=False)
        with torch.no_grad():
            wq, wk, wv = (params[f"{pre}/self_attention/{n}/kernel"].reshape(H, H) for n in ("query", "key", "value"))
            bq, bk, bv = (params[f"{pre}/self_attention/{n}/bias"].reshape(H) for n in ("query", "key", "value"))
            layer.self_attn.in_proj_weight.copy_(torch.cat([wq.t(), wk.t(), wv.t()], 0))     # torch: y = x W^T
            layer.self_attn.in_proj_bias.copy_(torch.cat([bq, bk, bv], 0))
            layer.self_attn.out_proj.weight.copy_(params[f"{pre}/self_attention/attention_output/kernel"].reshape(H, H).t())
            layer.self_attn.out_proj.bias.copy_(params[f"{pre}/self_attention/attention_output/bias"])
            layer.norm1.weight.copy_(params[f"{pre}/self_attention_layer_norm/gamma"])
            layer.norm1.bias.copy_(params[f"{pre}/self_attention_layer_norm/beta"])
            layer.linear1.weight.copy_(params[f"{pre}/intermediate/kernel"].t())
            layer.linear1.bias.copy_(params[f"{pre}/intermediate/bias"])
            layer.linear2.weight.copy_(params[f"{pre}/output/kernel"].t())
            layer.linear2.bias.copy_(params[f"{pre}/output/bias"])
            layer.norm2.weight.copy_(params[f"{pre}/output_layer_norm/gamma"])
            layer.norm2.bias.copy_(params[f"{pre}/output_layer_norm/beta"])
        layer.eval()
        with torch.no_grad():
            # key padding as the float adder Keras uses (a boolean mask would put -inf: same softmax wherever a key is valid)
            kpm = (1.0 - mask.float()) * -1e9
            x = layer(x, src_key_padding_mask=kpm)
        got = out["encoder_outputs"][i]
        assert float((x - got).abs().max()) < 2e-5, (i, float((x - got).abs().max()))
