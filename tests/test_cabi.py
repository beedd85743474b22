"""The C-ABI library loads without a GPU and exports exactly what include/b4r.h declares; host-only queries (layouts,
argument validation, error messages) work on CPU.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from bert4rec_amd import _lib
from bert4rec_amd.engine import make_model_config, param_table
from oracle import bert4rec_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "b4r.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(b4r_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/b4r.h but not exported"
    assert sorted(_lib.PROTOTYPES) == syms, "bert4rec_amd/_lib.py PROTOTYPES and include/b4r.h disagree"
    assert lib.b4r_version() == 100


def test_struct_sizes_match_the_header():
    assert C.sizeof(_lib.ModelConfig) == 36 and C.sizeof(_lib.AdamWConfig) == 48   # 9 x 4 bytes, padding, the decay-mask pointer
    assert C.sizeof(_lib.Batch) == 4 * 8 + 3 * 4 + 4     # padded to 8
    assert _lib.STATE_WORDS * 4 == 64


def test_parameter_layout_ml1m():
    lib = _lib.load()
    cfg = make_model_config(3709, 64, 2, 2, 200, 256, 0.2, 0.2)
    table = param_table(cfg)
    assert sum(e.rows * e.cols for e in table) == 358269           # trainable floats at C1
    assert lib.b4r_param_total_floats(C.byref(cfg)) == 358272      # + padding of the [3709] output bias
    n_decay = lib.b4r_param_decay_floats(C.byref(cfg))
    # names and shapes are the reference's Keras variables; decay flags follow adam_w_optimizer.py:154-168
    ocfg = orc.OracleConfig(vocab_size=3709)
    want = {n: s for n, s in orc.param_names_and_shapes(ocfg) if orc.is_trainable(n)}
    assert {e.name for e in table} == set(want)
    for e in table:
        size = 1
        for d in want[e.name]:
            size *= d
        assert e.rows * e.cols == size, e.name
        assert bool(e.decay) == orc.uses_weight_decay(e.name), e.name
        assert (e.offset < n_decay) == bool(e.decay), e.name
        assert e.offset % 4 == 0
    q = next(e for e in table if e.name.endswith("layer_1/self_attention/key/kernel"))
    assert (q.rows, q.cols, q.ld) == (64, 64, 192)                  # column block of the fused [H,3H] QKV matrix
    assert lib.b4r_pooler_floats(C.byref(cfg)) == 64 * 64 + 64


def test_workspace_queries_and_regions():
    lib = _lib.load()
    cfg = make_model_config(3709, 64, 2, 2, 200, 256)
    nbytes = lib.b4r_workspace_bytes(C.byref(cfg), 256, 200, 40)
    assert nbytes > 256 * 40 * 3709 * 4
    off, rows, cols, ld = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.b4r_workspace_region(C.byref(cfg), 256, 200, 40, b"mlm_logits", C.byref(off), C.byref(rows), C.byref(cols), C.byref(ld)) == 0
    assert (rows.value, cols.value, ld.value) == (10240, 3709, 3712) and off.value % 4 == 0
    assert lib.b4r_workspace_region(C.byref(cfg), 256, 200, 40, b"encoder_output_1", C.byref(off), C.byref(rows), C.byref(cols), C.byref(ld)) == 0
    assert (rows.value, cols.value) == (51200, 64)
    assert lib.b4r_workspace_region(C.byref(cfg), 256, 200, 40, b"nonsense", C.byref(off), C.byref(rows), C.byref(cols), C.byref(ld)) == -1
    assert "unknown region" in _lib.last_error()


@pytest.mark.parametrize("bad", [dict(hidden=96, heads=3), dict(hidden=64, heads=4), dict(hidden=2048, heads=64), dict(layers=0)])
def test_invalid_geometry_is_reported_not_crashed(bad):
    lib = _lib.load()
    cfg = make_model_config(100, bad.get("hidden", 64), bad.get("layers", 2), bad.get("heads", 2), 50, 256)
    assert lib.b4r_param_total_floats(C.byref(cfg)) == -1
    assert len(_lib.last_error()) > 0


def test_null_arguments_return_error_codes():
    lib = _lib.load()
    assert lib.b4r_gemm_f32(None, None) == -1 and "null" in _lib.last_error()
    assert lib.b4r_state_begin_step(None, None) == -1
    d = _lib.GemmDesc()
    d.A = d.B = d.C = 16
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = 4, 4, 8, 4, 4, 4     # lda < K
    assert lib.b4r_gemm_f32(C.byref(d), None) == -2
