"""GPU edge cases of the hot path against the oracle: minimum and maximum sequence lengths, single rows, slots that are
all ignored but one, the length limit, bad arguments.  Both arithmetic modes."""
import ctypes as C

import pytest
import torch

from bert4rec_amd import _lib
from bert4rec_amd.engine import Engine, make_model_config
from oracle import bert4rec_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]


def build(V, L, layers=1, heads=2, inner=64):
    cfg_o = orc.OracleConfig(vocab_size=V, hidden_size=32 * heads, num_layers=layers, num_attention_heads=heads,
                             max_sequence_length=L, inner_dim=inner)
    eng = Engine(make_model_config(V, 32 * heads, layers, heads, L, inner, 0.0, 0.0), "cuda")
    params = orc.init_params(cfg_o, 5)
    eng.load_named(params)
    return eng, params, cfg_o


def check(eng, params, cfg_o, batch, grads=True):
    loss_ref, grads_ref, out_ref = orc.loss_and_grads(params, batch, cfg_o, training=False)
    cb, keep = eng.prepare_batch(batch)
    eng.begin_step()
    eng.forward(cb, training=False, pooler=True)
    B, L, P = cb.B, cb.L, cb.P
    logits = eng.region("mlm_logits", B, L, P).view(B, P, -1).cpu()
    assert float((logits - out_ref["mlm_logits"]).abs().max()) < 1e-3
    assert float((eng.region("pooled_output", B, L, P).cpu() - out_ref["pooled_output"]).abs().max()) < 1e-3
    if grads:
        eng.loss(cb, want_grad=True)
        eng.backward(cb, training=False)
        torch.cuda.synchronize()
        st = eng.read_state()
        assert abs(st["loss_sum"] / st["valid_count"] - float(loss_ref)) < 1e-3
        got = eng.export_named(eng.grads)
        floor = 1e-4 * max(float(g.abs().max()) for g in grads_ref.values()) + 1e-7   # key bias: analytically zero
        for n, g in grads_ref.items():
            a = got[n].double() / st["valid_count"]
            b = g.double().reshape(a.shape)
            assert float((a - b).abs().max()) <= 2e-3 * max(float(b.abs().max()), floor), n


@pytest.mark.parametrize("B,L,P", [(1, 1, 1), (1, 2, 1), (2, 3, 2), (1, 64, 4), (2, 65, 3), (1, 256, 8), (32, 16, 4)])
def test_sequence_length_extremes(B, L, P):
    eng, params, cfg_o = build(53, L)
    batch = orc.synthetic_batch(B, L, P, 53, seed=B + L, ragged=(L > 4), rate=0.5)
    check(eng, params, cfg_o, batch)


def test_rows_with_a_single_valid_slot_and_repeated_padding_positions():
    """finetune / validation rows: slot 0 valid, every other slot is (position 0, id 0) (bert4rec_preprocessor.py:95-99)"""
    eng, params, cfg_o = build(71, 24, layers=2)
    batch = orc.synthetic_batch(6, 24, 5, 71, seed=3, ragged=True, finetune=True)
    assert int((batch["masked_lm_ids"] != 0).sum()) == 6 and int(batch["masked_lm_positions"][:, 1:].abs().sum()) == 0
    check(eng, params, cfg_o, batch)


def test_masked_positions_may_repeat():
    """the kernels scatter-add: duplicated positions (not produced by the reference's preprocessor) still sum correctly"""
    eng, params, cfg_o = build(40, 12)
    batch = orc.synthetic_batch(3, 12, 4, 40, seed=9)
    batch["masked_lm_positions"][0] = torch.tensor([5, 5, 5, 7])
    batch["masked_lm_ids"][0] = torch.tensor([9, 9, 11, 12])
    check(eng, params, cfg_o, batch)


def test_limits_are_reported():
    lib = _lib.load()
    eng, params, cfg_o = build(30, 300)
    batch = orc.synthetic_batch(1, 257, 2, 30, seed=1)
    cb, keep = eng.prepare_batch(batch)
    with pytest.raises(_lib.B4RError, match="256"):
        eng.forward(cb)
    eng2, _, _ = build(30, 16)
    long_batch = orc.synthetic_batch(2, 20, 2, 30, seed=1)
    cb2, keep2 = eng2.prepare_batch(long_batch)
    with pytest.raises(_lib.B4RError, match="max_sequence_length"):
        eng2.forward(cb2)
    with pytest.raises(ValueError):
        eng2.prepare_batch({"input_word_ids": torch.zeros(2, 4, dtype=torch.int64)})
    with pytest.raises(ValueError):
        eng2.prepare_batch({"input_word_ids": torch.zeros(4, dtype=torch.int64), "input_mask": torch.zeros(4, dtype=torch.int64)})
    # workspace too small is an error code, not a fault
    cb3, keep3 = eng2.prepare_batch(orc.synthetic_batch(2, 16, 2, 30, seed=2))
    small = torch.empty(1024, device="cuda")
    rc = lib.b4r_forward(C.byref(eng2.cfg), C.byref(cb3), eng2.params.data_ptr(), eng2.pooler.data_ptr(), small.data_ptr(),
                         small.numel() * 4, eng2.state.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    assert rc == -5 and "workspace too small" in _lib.last_error()


def test_out_of_range_ids_read_the_pad_row_instead_of_faulting():
    eng, params, cfg_o = build(30, 8)
    batch = orc.synthetic_batch(2, 8, 2, 30, seed=4)
    bad = {k: v.clone() for k, v in batch.items()}
    bad["input_word_ids"][0, 0] = 999
    ref = {k: v.clone() for k, v in batch.items()}
    ref["input_word_ids"][0, 0] = 0
    cb, keep = eng.prepare_batch(bad)
    eng.forward(cb)
    got = eng.region("mlm_logits", 2, 8, 2).view(2, 2, -1).cpu().clone()
    want = orc.model_forward(params, ref, cfg_o)["mlm_logits"]
    assert float((got - want).abs().max()) < 1e-3
