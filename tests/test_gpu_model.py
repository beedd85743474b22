"""GPU parity tests, model level: forward / loss / gradients / optimizer steps of the HIP path against the oracle
(oracle/bert4rec_oracle.py, torch-CPU fp32 + autograd) on the same seeded inputs and weights.

Tolerance (BASELINE.json north_star): logits and loss within 1e-3 in fp32.  The fp32 matrix-core path is expected to sit
near 1e-5; the asserts use 1e-3 for logits/loss and a relative 2e-3 for gradients (fp32 accumulation order differs)."""
import numpy as np
import pytest
import torch

from bert4rec_amd import _lib
from bert4rec_amd.engine import Engine, make_adamw_config, make_model_config
from oracle import bert4rec_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]

LOGIT_TOL = 1e-3


def build(cfg_o: orc.OracleConfig, seed=3):
    cfg = make_model_config(cfg_o.vocab_size, cfg_o.hidden_size, cfg_o.num_layers, cfg_o.num_attention_heads,
                            cfg_o.max_sequence_length, cfg_o.inner_dim, cfg_o.output_dropout, cfg_o.attention_dropout)
    eng = Engine(cfg, "cuda")
    params = orc.init_params(cfg_o, seed)
    # make biases / LN parameters non-trivial so that their gradients and uses are exercised
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in params.items():
        if n.endswith(("bias", "beta")):
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        elif n.endswith("gamma"):
            p.copy_(1.0 + torch.randn(p.shape, generator=g) * 0.05)
    eng.load_named(params)
    return eng, params


def outputs(eng, cb):
    B, L, P = cb.B, cb.L, cb.P
    nl = eng.cfg.num_layers
    out = {"sequence_output": eng.region("sequence_output", B, L, P).view(B, L, -1),
           "pooled_output": eng.region("pooled_output", B, L, P),
           "encoder_outputs": [eng.region(f"encoder_output_{i}", B, L, P).view(B, L, -1) for i in range(nl)]}
    if P > 0:
        out["mlm_logits"] = eng.region("mlm_logits", B, L, P).view(B, P, -1)
        out["mlm_hidden"] = eng.region("mlm_hidden", B, L, P).view(B, P, -1)
    return out


CONFIGS = {
    "tiny": (orc.OracleConfig(vocab_size=37, hidden_size=64, num_layers=2, num_attention_heads=2, max_sequence_length=20,
                              inner_dim=64), dict(B=4, L=16, P=5)),
    "ml1m_slice": (orc.OracleConfig(vocab_size=3709, hidden_size=64, num_layers=2, num_attention_heads=2,
                                    max_sequence_length=200, inner_dim=256), dict(B=8, L=200, P=40)),
    "h128": (orc.OracleConfig(vocab_size=500, hidden_size=128, num_layers=1, num_attention_heads=4, max_sequence_length=50,
                              inner_dim=512), dict(B=8, L=48, P=20)),
    "h256": (orc.OracleConfig(vocab_size=1000, hidden_size=256, num_layers=2, num_attention_heads=8, max_sequence_length=64,
                              inner_dim=1024), dict(B=4, L=40, P=8)),
    # sequences of more than two 32-token tiles with at most 64 slots: under B4R_FLAG_HEAD_ROWS_ONLY the last layer's attention core runs
    # with the slots as its only queries (one / two compact query tiles: P = 24 / 40)
    "h128_long": (orc.OracleConfig(vocab_size=700, hidden_size=128, num_layers=2, num_attention_heads=4, max_sequence_length=100,
                                   inner_dim=512), dict(B=6, L=96, P=24)),
    "h256_long": (orc.OracleConfig(vocab_size=900, hidden_size=256, num_layers=1, num_attention_heads=8, max_sequence_length=200,
                                   inner_dim=1024), dict(B=3, L=200, P=40)),
    # token / slot counts that are NOT multiples of 32: the dense layers fall back to the exact-fp32 LDS-tiled kernels
    "odd_rows": (orc.OracleConfig(vocab_size=301, hidden_size=64, num_layers=1, num_attention_heads=2, max_sequence_length=50,
                                  inner_dim=256), dict(B=5, L=50, P=7)),
}


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.mark.parametrize("name", list(CONFIGS))
@pytest.mark.parametrize("ragged", [False, True])
def test_forward_matches_oracle(name, ragged):
    cfg_o, shp = CONFIGS[name]
    eng, params = build(cfg_o)
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=1, ragged=ragged)
    ref = orc.model_forward(params, batch, cfg_o, training=False)
    cb, keep = eng.prepare_batch(batch)
    eng.forward(cb, training=False, pooler=True)
    got = outputs(eng, cb)
    assert maxdiff(got["sequence_output"], ref["sequence_output"]) < LOGIT_TOL
    for a, b in zip(got["encoder_outputs"], ref["encoder_outputs"]):
        assert maxdiff(a, b) < LOGIT_TOL
    assert maxdiff(got["pooled_output"], ref["pooled_output"]) < LOGIT_TOL
    assert maxdiff(got["mlm_hidden"], ref["mlm_hidden"]) < LOGIT_TOL
    d = maxdiff(got["mlm_logits"], ref["mlm_logits"])
    assert d < LOGIT_TOL, d
    # expected to be far inside the tolerance with exact-fp32 matrix cores
    assert d < 2e-4, d


def run_loss_and_grads(eng, batch, training, seed=0, step=0, fused_head=False, head_rows_only=None):
    """fused_head=True runs what b4r_train_step runs: the logits-free head AND the last layer's feed-forward half restricted to the
    rows the head gathers (B4R_FLAG_HEAD_ROWS_ONLY); False the materialising head on all rows."""
    cb, keep = eng.prepare_batch(batch)
    eng.set_seed(seed)
    eng.set_step(step)
    eng.begin_step()
    rows = fused_head if head_rows_only is None else head_rows_only
    eng.forward(cb, training=training, pooler=False, fused_head=fused_head, head_rows_only=rows)
    eng.loss(cb, want_grad=True, fused_head=fused_head)
    eng.backward(cb, training=training, fused_head=fused_head, head_rows_only=rows)
    torch.cuda.synchronize()
    st = eng.read_state()
    return st, eng.export_named(eng.grads)


def compare_grads(got, ref, count, rel=2e-3):
    worst = ("", 0.0)
    # the key-bias gradient is analytically zero (softmax is shift invariant), so its reference is rounding noise: every
    # tensor is measured against at least 1e-4 of the largest gradient magnitude in the model
    floor = 1e-4 * max(float(g.abs().max()) for g in ref.values()) + 1e-7
    for n, g in ref.items():
        a = got[n].double() / count
        b = g.double().reshape(a.shape)
        scale = max(float(b.abs().max()), floor)
        err = float((a - b).abs().max()) / scale
        if err > worst[1]:
            worst = (n, err)
        assert err < rel, f"gradient of {n}: relative error {err:.3e} (scale {scale:.3e})"
    return worst


@pytest.mark.parametrize("name", list(CONFIGS))
def test_loss_and_gradients_match_autograd(name):
    cfg_o, shp = CONFIGS[name]
    eng, params = build(cfg_o)
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=2, ragged=True)
    loss_ref, grads_ref, _ = orc.loss_and_grads(params, batch, cfg_o, training=False)
    st, grads = run_loss_and_grads(eng, batch, training=False)
    assert st["valid_count"] == float((batch["masked_lm_ids"] != 0).sum())
    assert abs(st["loss_sum"] / st["valid_count"] - float(loss_ref)) < LOGIT_TOL
    compare_grads(grads, grads_ref, st["valid_count"])
    if eng.fused_head_supported():
        # the train-step head that never materialises the logits: same loss, metrics and gradients
        st2, grads2 = run_loss_and_grads(eng, batch, training=False, fused_head=True)
        assert st2["valid_count"] == st["valid_count"]
        assert abs(st2["loss_sum"] / st2["valid_count"] - float(loss_ref)) < LOGIT_TOL
        assert st2["correct_masked"] == st["correct_masked"] and st2["correct_all"] == st["correct_all"]
        compare_grads(grads2, grads_ref, st2["valid_count"])


def test_train_mode_with_dropout_matches_oracle_mask_for_mask():
    """The counter-hash dropout is restated in the oracle, so train-mode loss and gradients are comparable too."""
    cfg_o, shp = CONFIGS["tiny"]
    cfg_o = orc.OracleConfig(**{**cfg_o.__dict__, "output_dropout": 0.2, "attention_dropout": 0.2})
    eng, params = build(cfg_o)
    batch = orc.synthetic_batch(6, 16, 5, cfg_o.vocab_size, seed=3, ragged=True)
    seed, step = 4242, 17
    loss_ref, grads_ref, _ = orc.loss_and_grads(params, batch, cfg_o, training=True, rng=(seed, step))
    st, grads = run_loss_and_grads(eng, batch, training=True, seed=seed, step=step)
    assert abs(st["loss_sum"] / st["valid_count"] - float(loss_ref)) < LOGIT_TOL
    compare_grads(grads, grads_ref, st["valid_count"], rel=5e-3)
    # and the masks really were applied: eval-mode loss differs
    loss_eval, _, _ = orc.loss_and_grads(params, batch, cfg_o, training=False)
    assert abs(float(loss_eval) - float(loss_ref)) > 1e-4


def test_train_mode_at_sequence_length_216_matches_oracle_mask_for_mask():
    """L = 216 at hidden size 64: the forward runs the 32-token-tile attention block (it writes the attention-dropout decisions in the
    32-key-tile layout only), the backward is past the resident block's limit (L > 208) and goes through the attention CORE kernels --
    which must read those decisions in the layout the forward wrote (one predicate decides both: b4r_attn_block_fwd only takes the
    32-token-tile kernel where the core backward reads its layout).  Wrong decision words show up as wrong gradients here."""
    cfg_o = orc.OracleConfig(vocab_size=211, hidden_size=64, num_layers=1, num_attention_heads=2, max_sequence_length=216, inner_dim=256,
                             output_dropout=0.1, attention_dropout=0.3)
    eng, params = build(cfg_o)
    batch = orc.synthetic_batch(3, 216, 12, cfg_o.vocab_size, seed=5, ragged=True)
    batch["input_mask"][0] = 1                                      # one sequence of the full length
    seed, step = 99, 3
    loss_ref, grads_ref, _ = orc.loss_and_grads(params, batch, cfg_o, training=True, rng=(seed, step))
    st, grads = run_loss_and_grads(eng, batch, training=True, seed=seed, step=step)
    assert abs(st["loss_sum"] / st["valid_count"] - float(loss_ref)) < LOGIT_TOL
    compare_grads(grads, grads_ref, st["valid_count"], rel=5e-3)


def test_train_steps_follow_the_reference_optimizer():
    """k identical steps (bert4rec_model.py:151-173): same loss trajectory and same weights afterwards."""
    cfg_o, shp = CONFIGS["tiny"]
    eng, params = build(cfg_o)
    hp_o = orc.AdamWConfig(num_warmup_steps=2, num_train_steps=50)  # leave the lr=0 region quickly
    hp = make_adamw_config(hp_o.init_lr, hp_o.num_train_steps, hp_o.num_warmup_steps, hp_o.end_lr, hp_o.weight_decay_rate,
                           hp_o.beta_1, hp_o.beta_2, hp_o.epsilon, hp_o.gradient_clip_norm)
    m, v = orc.zeros_like_params(params), orc.zeros_like_params(params)
    batches = [orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=10 + i, ragged=True) for i in range(4)]
    eng.set_step(0)
    for i, batch in enumerate(batches):
        ref = orc.train_step(params, m, v, batch, cfg_o, hp_o, step=i, training=False)
        cb, keep = eng.prepare_batch(batch)
        eng.cfg.output_dropout = 0.0
        eng.train_step(hp, cb)
        torch.cuda.synchronize()
        st = eng.read_state()
        assert st["step"] == i + 1
        assert abs(st["loss_sum"] / st["valid_count"] - ref["loss"]) < LOGIT_TOL
        assert abs(st["grad_norm"] - ref["grad_norm"]) < 2e-3 * max(1.0, ref["grad_norm"])
        assert abs(st["correct_masked"] / st["valid_count"] - ref["masked_accuracy"]) < 1e-6
        assert abs(st["correct_all"] / st["slots_all"] - ref["sparse_categorical_accuracy"]) < 1e-6
    got = eng.export_named()
    for n, p in params.items():
        if orc.is_trainable(n):
            assert maxdiff(got[n], p.reshape(got[n].shape)) < 5e-6, n


def test_gradient_of_unused_rows_is_zero_and_pooler_untouched():
    cfg_o, shp = CONFIGS["tiny"]
    eng, params = build(cfg_o)
    batch = orc.synthetic_batch(4, 10, 5, cfg_o.vocab_size, seed=5)
    st, grads = run_loss_and_grads(eng, batch, training=False)
    gpos = grads["position_embedding/embeddings"]
    assert float(gpos[10:].abs().max()) == 0.0 and float(gpos[:10].abs().max()) > 0.0
    assert "pooler_transform/kernel" not in grads


def test_graph_replayed_steps_equal_eager_steps():
    """Engine.train_step_graphed: the step-varying scalars live in the device state, so replaying a captured hipGraph is a
    new train step each time (eager, capture + replay, replay, ... must follow the eager trajectory)."""
    cfg_o, shp = CONFIGS["tiny"]
    cfg_o = orc.OracleConfig(**{**cfg_o.__dict__, "output_dropout": 0.1, "attention_dropout": 0.1})
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=8, ragged=True)
    from bert4rec_amd.engine import make_adamw_config
    hp = make_adamw_config(num_warmup_steps=2, num_train_steps=20)
    runs = []
    for graphed in (False, True):
        eng, _ = build(cfg_o)
        eng.set_seed(77)
        cb, keep = eng.prepare_batch(batch)
        losses = []
        for it in range(6):
            (eng.train_step_graphed if graphed else eng.train_step)(hp, cb)
            torch.cuda.synchronize()
            # other batch shapes between the replays (evaluation, ranking, a last partial batch): more workspaces than the
            # engine caches.  The workspace a captured graph points into must survive, and be left alone by the allocator
            for k in range(6):
                other = orc.synthetic_batch(2 + k, 8 + 2 * it % 6, 3, cfg_o.vocab_size, seed=k)
                ocb, okeep = eng.prepare_batch(other)
                eng.forward(ocb, training=False, pooler=False)
                torch.empty(1 << 20, device="cuda").fill_(float("nan"))   # would land in a freed workspace
            torch.cuda.synchronize()
            st = eng.read_state()
            losses.append((st["step"], st["loss_sum"], st["lr"]))
        runs.append((losses, eng.params.cpu().clone()))
    (l0, p0), (l1, p1) = runs
    assert [s for s, _, _ in l1] == [1, 2, 3, 4, 5, 6] == [s for s, _, _ in l0]
    for (s0, a, lra), (s1, b, lrb) in zip(l0, l1):
        assert abs(a - b) <= 1e-4 * abs(a) and lra == lrb
    assert float((p0 - p1).abs().max()) < 1e-6


def test_backward_can_form_the_loss_sums_itself_bit_for_bit():
    """B4R_FLAG_LOSS_SUMS: b4r_backward sets the state's loss / metric sums inside its first launch, in b4r_loss's summation order:
    the same bits as begin_step + b4r_loss(fused head) + b4r_backward, gradients included (what b4r_train_step and the
    data-parallel step run)."""
    cfg_o, shp = CONFIGS["ml1m_slice"]
    eng, _ = build(cfg_o)
    if not eng.fused_head_supported():
        pytest.skip("the logits-free head needs the bf16x3 mode")
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=4, ragged=True)
    st_a, g_a = run_loss_and_grads(eng, batch, training=True, seed=5, step=2, fused_head=True)
    cb, keep = eng.prepare_batch(batch)
    eng.set_seed(5)
    eng.set_step(2)
    eng.state[4:11] = 12345           # stale sums: the flag overwrites, it does not add
    eng.forward(cb, training=True, pooler=False, fused_head=True, head_rows_only=True)
    eng.backward(cb, training=True, fused_head=True, head_rows_only=True, loss_sums=True, grad_tail=True)
    torch.cuda.synchronize()
    st_b, g_b = eng.read_state(), eng.export_named(eng.grads)
    for k in ("loss_sum", "valid_count", "correct_masked", "correct_all", "slots_all"):
        assert st_a[k] == st_b[k], k
    for n in g_a:      # the item table too: its scatter sums in 64-bit fixed point, order-free
        assert torch.equal(g_a[n], g_b[n]), n
    tail = eng.grad_ext[eng.n_params:eng.n_params + 5].cpu().tolist()
    assert tail == [st_b[k] for k in ("loss_sum", "valid_count", "correct_masked", "correct_all", "slots_all")]


def test_train_steps_are_bitwise_reproducible():
    """Two runs of 50 train steps (dropout on, ragged batch, many tokens per item row so that the item-table scatter has real
    contention) from the same weights, seed and batch: every parameter and every step's loss sum identical to the bit.  The one
    order-dependent sum of a step -- the item-table scatter-add -- is formed in 64-bit fixed point (b4r_rowops.hip)."""
    cfg_o, shp = CONFIGS["ml1m_slice"]
    cfg_o = orc.OracleConfig(**{**cfg_o.__dict__, "vocab_size": 301, "output_dropout": 0.2, "attention_dropout": 0.2})
    batch = orc.synthetic_batch(32, shp["L"], shp["P"], cfg_o.vocab_size, seed=21, ragged=True)
    hp = make_adamw_config(num_warmup_steps=5, num_train_steps=100)
    runs = []
    for rep in range(2):
        eng, _ = build(cfg_o)
        eng.set_seed(1234)
        cb, keep = eng.prepare_batch(batch)
        losses = []
        for it in range(50):
            eng.train_step(hp, cb)
            if it % 10 == 9:
                torch.cuda.synchronize()
                losses.append(eng.read_state()["loss_sum"])
            if rep == 1 and it % 7 == 0:      # perturb the timing between the steps of the second run
                torch.randn(1 << 18, device="cuda").sum().item()
        torch.cuda.synchronize()
        runs.append((losses, eng.params.clone(), eng.grads.clone()))
    (l0, p0, g0), (l1, p1, g1) = runs
    assert l0 == l1
    assert torch.equal(g0, g1) and torch.equal(p0, p1)
    assert np.isfinite(l0[-1]) and l0[-1] != l0[0]


def test_train_step_takes_the_gradient_norm_from_the_closing_reduce_launch():
    """b4r_train_step has no norm launch when the backward's closing reduce launch writes every gradient: it squares each value as
    it stores it.  The norm the optimizer used (state) must be the norm of the gradient buffer the step leaves behind, and the step
    must have one launch fewer than forward + backward + b4r_optimizer_step."""
    import ctypes as C
    cfg_o, shp = CONFIGS["ml1m_slice"]
    cfg_o = orc.OracleConfig(**{**cfg_o.__dict__, "output_dropout": 0.1, "attention_dropout": 0.1})
    eng, _ = build(cfg_o)
    if not eng.fused_head_supported():
        pytest.skip("the closing reduce launch covers every gradient in the bf16x3 step only")
    hp = make_adamw_config(num_warmup_steps=2, num_train_steps=20, gradient_clip_norm=0.05)   # small: the clip is active
    cb, keep = eng.prepare_batch(orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=9, ragged=True))
    eng.set_seed(3)
    for _ in range(3):
        eng.train_step(hp, cb)
    torch.cuda.synchronize()
    st = eng.read_state()
    g = eng.grads[:eng.n_params].double()
    want = float(g.pow(2).sum().sqrt()) / st["valid_count"]
    assert want > 0.0 and abs(st["grad_norm"] - want) <= 2e-6 * want
    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    n = C.c_int32(0)
    us = (C.c_float * 256)()
    names = C.create_string_buffer(256 * 128)
    _lib.check(lib.b4r_timing_begin(stream, 256), "b4r_timing_begin")
    eng.train_step(hp, cb)
    _lib.check(lib.b4r_timing_end(C.byref(n), us, names, 128, 256), "b4r_timing_end")
    labels = [names.raw[j * 128:(j + 1) * 128].split(b"\0", 1)[0].decode() for j in range(n.value)]
    assert "global norm" not in labels and labels.count("multi_slab_reduce") == 1, labels


@pytest.mark.parametrize("name", ["ml1m_slice", "h128", "h256", "h128_long", "h256_long"])
def test_encoder_only_forward_on_the_ranked_rows_equals_the_full_forward_there(name):
    """B4R_FLAG_ENCODER_ONLY | B4R_FLAG_HEAD_ROWS_ONLY (what an evaluation runs): no masked-LM head although the batch carries
    masked_lm_positions / masked_lm_ids, the last layer's feed-forward half only on the rows of the valid slots.  Those rows of the
    sequence output must equal the full forward's within the products' rounding (the last layer sweeps only those queries, with the
    softmax of a query merged from per-key-tile partials -- another summation order; at 128 / 256 the encoder-only forward also runs the
    one-launch feed-forward block of b4r_ffn32w.hip in every layer but the last and the last layer's products on the gathered rows);
    the logits region must stay untouched."""
    cfg_o, shp = CONFIGS[name]
    eng, _ = build(cfg_o)
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=12, ragged=True)
    cb, keep = eng.prepare_batch(batch)
    B, L, P = cb.B, cb.L, cb.P
    eng.forward(cb, training=False, pooler=False)
    torch.cuda.synchronize()
    full = eng.region("sequence_output", B, L, P).clone()
    eng.region("sequence_output", B, L, P).fill_(float("nan"))
    eng.region("mlm_logits", B, L, P).fill_(7.0)
    eng.forward(cb, training=False, pooler=False, head_rows_only=True, encoder_only=True)
    torch.cuda.synchronize()
    got = eng.region("sequence_output", B, L, P)
    valid = (batch["masked_lm_ids"] != 0)
    rows = (torch.arange(B)[:, None] * L + batch["masked_lm_positions"].clamp(0, L - 1))[valid].to(got.device)
    assert rows.numel() > 0
    assert maxdiff(got[rows], full[rows]) < 2e-5
    assert bool((eng.region("mlm_logits", B, L, P) == 7.0).all())
    if eng.fused_head_supported():   # (the fused feed-forward block: only there are the other rows skipped)
        others = torch.ones(B * L, dtype=torch.bool, device=got.device)
        others[rows] = False
        skipped = torch.isnan(got[others]).all(dim=1).float().mean().item()   # (rows of padded slots may be written too: harmless)
        assert skipped > 0.7, skipped


@pytest.mark.parametrize("name", ["h128", "h256", "h128_long", "h256_long"])
def test_wide_train_mode_on_the_heads_rows_matches_oracle_mask_for_mask(name):
    """Hidden sizes 128 / 256 in train mode with both dropouts on, as b4r_train_step runs them: the logits-free head and the last
    layer's feed-forward half on the gathered rows only (compact [B*P, .] operands; the dropout decisions are those of the sequence
    rows).  Loss and every gradient against the oracle's autograd with the same masks; and the dense form of the same step (flag
    off) gives the same gradients."""
    cfg_o, shp = CONFIGS[name]
    cfg_o = orc.OracleConfig(**{**cfg_o.__dict__, "output_dropout": 0.2, "attention_dropout": 0.2})
    eng, params = build(cfg_o)
    fused = eng.fused_head_supported()    # (exact-fp32 mode: the materialising head, the rows mode all the same)
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=5, ragged=True)
    seed, step = 977, 4
    loss_ref, grads_ref, _ = orc.loss_and_grads(params, batch, cfg_o, training=True, rng=(seed, step))
    st, grads = run_loss_and_grads(eng, batch, training=True, seed=seed, step=step, fused_head=fused, head_rows_only=True)
    assert abs(st["loss_sum"] / st["valid_count"] - float(loss_ref)) < LOGIT_TOL
    compare_grads(grads, grads_ref, st["valid_count"], rel=5e-3)
    grads = {k: v.clone() for k, v in grads.items()}
    st2, dense = run_loss_and_grads(eng, batch, training=True, seed=seed, step=step, fused_head=fused, head_rows_only=False)
    assert abs(st2["loss_sum"] - st["loss_sum"]) < 1e-5 * abs(st["loss_sum"])
    floor = 1e-4 * max(float(g.abs().max()) for g in dense.values())
    for k, g in dense.items():
        assert float((grads[k] - g).abs().max()) < 2e-4 * max(float(g.abs().max()), floor), k


def test_slot_query_attention_with_a_fully_masked_sequence_equals_the_dense_path():
    """The last layer's attention with the slots as its only queries (hidden 128, L = 96, P = 24) on a batch whose first sequence has
    EVERY key masked (Keras' -1e9 adder then gives a uniform softmax over all L keys) and whose second has a single real token: loss
    and gradients equal the dense path's (flag off), which the block tests pin against fp64 autograd for these cases."""
    cfg_o, shp = CONFIGS["h128_long"]
    cfg_o = orc.OracleConfig(**{**cfg_o.__dict__, "output_dropout": 0.1, "attention_dropout": 0.3})
    eng, params = build(cfg_o)
    if not eng.fused_head_supported():
        pytest.skip("the compact last layer runs in the bf16x3 mode")
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=8, ragged=True)
    batch["input_mask"][0, :] = 0
    batch["input_mask"][1, 1:] = 0
    st, grads = run_loss_and_grads(eng, batch, training=True, seed=31, step=2, fused_head=True, head_rows_only=True)
    grads = {k: v.clone() for k, v in grads.items()}
    st2, dense = run_loss_and_grads(eng, batch, training=True, seed=31, step=2, fused_head=True, head_rows_only=False)
    assert st["valid_count"] == st2["valid_count"] and abs(st2["loss_sum"] - st["loss_sum"]) < 1e-5 * abs(st["loss_sum"])
    floor = 1e-4 * max(float(g.abs().max()) for g in dense.values())
    for k, g in dense.items():
        assert bool(torch.isfinite(grads[k]).all()), k
        assert float((grads[k] - g).abs().max()) < 2e-4 * max(float(g.abs().max()), floor), k


def test_item_table_gradient_with_hundreds_of_contributions_per_row():
    """The fixed-point item-table sum against autograd on a batch where every item row receives ten or more contributions
    (64 sequences over 37 items; the PAD and [MASK] rows hundreds): same tolerance as the other gradients, and the gradient of an
    item that does not occur in the batch is exactly the head's part."""
    cfg_o, shp = CONFIGS["tiny"]
    eng, params = build(cfg_o)
    batch = orc.synthetic_batch(64, shp["L"], shp["P"], cfg_o.vocab_size, seed=2, ragged=True)
    _, grads_ref, _ = orc.loss_and_grads(params, batch, cfg_o, training=False)
    st, grads = run_loss_and_grads(eng, batch, training=False)
    a = grads["word_embeddings/embeddings"].double() / st["valid_count"]
    b = grads_ref["word_embeddings/embeddings"].double()
    assert float((a - b).abs().max()) < 2e-3 * float(b.abs().max())
    counts = torch.bincount(batch["input_word_ids"].reshape(-1), minlength=cfg_o.vocab_size)
    assert int(counts.max()) > 300 and int((counts >= 10).sum()) > 30


def test_item_table_scatter_reports_out_of_range_contributions_instead_of_wrapping():
    """The item-table scatter adds in 64-bit fixed point (units of 2^-36: a 64-bit sum holds 2^27).  A contribution that is not finite or
    reaches 2^18 in magnitude cannot be represented: float atomics would have carried it (or an Inf / NaN) into the gradient, round 3's
    integer sums wrapped silently into finite garbage.  Now a sticky poison word makes the closing reduction store NaN for the whole
    table gradient.  The backward is linear in d loss / d logits, so scaling that tensor between b4r_loss and b4r_backward scales
    every contribution: x 1 must reproduce the oracle, x 1e9 (contributions of ~1e7: finite in fp32, out of range here) must poison
    the table gradient and leave the gradients that do not pass through the scatter finite."""
    cfg_o, shp = CONFIGS["tiny"]
    batch = orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=2, ragged=True)
    for scale in (1.0, 1e9):
        eng, params = build(cfg_o)
        cb, keep = eng.prepare_batch(batch)
        eng.begin_step()
        eng.forward(cb, training=False, pooler=False)
        eng.loss(cb, want_grad=True)
        eng.region("mlm_logits", cb.B, cb.L, cb.P).mul_(scale)
        eng.backward(cb, training=False)
        torch.cuda.synchronize()
        g = eng.export_named(eng.grads)
        table, other = g["word_embeddings/embeddings"], g["transformer/layer_0/intermediate/kernel"]
        assert bool(torch.isfinite(other).all()) and float(other.abs().max()) > 0
        if scale == 1.0:
            _, ref, _ = orc.loss_and_grads(params, batch, cfg_o, training=False)
            cnt = float((batch["masked_lm_ids"] != 0).sum())
            want = ref["word_embeddings/embeddings"] * cnt
            assert maxdiff(table, want) < 2e-3 * float(want.abs().max())
        else:
            assert bool(torch.isnan(table).all()), "out-of-range contributions must poison the table gradient, not wrap"


def test_launch_timer_lists_the_launches_of_a_train_step_in_order():
    """b4r_timing_begin / b4r_timing_end (what bench.py builds its per-step breakdown and its roofline block from): one entry per
    launch in enqueue order, the same sequence for every step, positive durations that add up to about the step's GPU time."""
    import ctypes as C
    from bert4rec_amd import _lib
    from bert4rec_amd.engine import make_adamw_config
    cfg_o, shp = CONFIGS["ml1m_slice"]
    eng, _ = build(cfg_o)
    if not eng.fused_head_supported():
        pytest.skip("counts the launches of the bf16x3 step")
    lib = _lib.load()
    hp = make_adamw_config()
    cb, keep = eng.prepare_batch(orc.synthetic_batch(shp["B"], shp["L"], shp["P"], cfg_o.vocab_size, seed=3))
    for _ in range(3):
        eng.train_step(hp, cb)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    cap, stride, steps = 512, 128, 3
    _lib.check(lib.b4r_timing_begin(stream, cap), "b4r_timing_begin")
    for _ in range(steps):
        eng.train_step(hp, cb)
    n = C.c_int32(0)
    us = (C.c_float * cap)()
    names = C.create_string_buffer(cap * stride)
    _lib.check(lib.b4r_timing_end(C.byref(n), us, names, stride, cap), "b4r_timing_end")
    assert n.value % steps == 0 and 15 <= n.value // steps <= 40
    per = n.value // steps
    labels = [names.raw[j * stride:(j + 1) * stride].split(b"\\0", 1)[0].decode() for j in range(n.value)]
    assert labels[:per] == labels[per:2 * per] == labels[2 * per:]
    assert any(l.startswith("b4r_attn_block_bwd") for l in labels[:per]) and any("head" in l for l in labels[:per])
    assert all(us[j] > 0.0 for j in range(n.value)) and 50.0 < sum(us[j] for j in range(per)) < 5000.0
    # outside a begin / end pair nothing is recorded
    eng.train_step(hp, cb)
    n2 = C.c_int32(-1)
    assert lib.b4r_timing_end(C.byref(n2), us, names, stride, cap) != 0 or n2.value == 0
