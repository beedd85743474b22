"""GPU parity tests, op level: every HIP kernel is called through the C ABI and compared with a plain torch-CPU fp32/fp64
restatement of the same op (tolerances stated per test).  Dropout masks are compared exactly through the oracle's
restatement of the counter hash."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from bert4rec_amd import _lib
from oracle import bert4rec_oracle as orc
from tests import b4r_testlib as T
from tests.b4r_testlib import P, stream

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_mode")]

DEV = "cuda"
TOL = 2e-5  # fp32 matrix-core sums of <= 4k terms of O(1) magnitude vs fp64


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float32)


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("M,N,K", [(300, 70, 64), (128, 64, 32), (257, 192, 100), (64, 256, 256), (5, 3, 7), (512, 192, 64),
                                   (96, 256, 64), (1024, 64, 64), (256, 64, 256), (128, 64, 192), (160, 128, 512), (64, 100, 128),
                                   (384, 320, 256), (96, 1024, 128), (288, 132, 160)])   # the last three: 128 x 128 LDS tiles
def test_gemm_nn_epilogues(M, N, K):
    A, B, bias, R = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.3), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = A.double() @ B.double()
    Ad, Bd, bd, Rd = A.to(DEV), B.to(DEV), bias.to(DEV), R.to(DEV)
    scale = max(1.0, float(ref.abs().max()))
    c, _ = T.gemm(Ad, Bd, M, N, K)
    assert T.maxdiff(c, ref) < TOL * scale
    c, _ = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_BIAS, bias=bd)
    assert T.maxdiff(c, ref + bias.double()) < TOL * scale
    qc = N // 3
    c, _ = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_BIAS_QSCALE, bias=bd, qscale=0.25, qcols=qc)
    want = ref + bias.double()
    want[:, :qc] *= 0.25
    assert T.maxdiff(c, want) < TOL * scale
    c, c2 = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_BIAS_GELU, bias=bd, want_c2=True)
    assert T.maxdiff(c2, ref + bias.double()) < TOL * scale
    assert T.maxdiff(c, gelu(ref + bias.double())) < TOL * scale
    c, _ = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_BIAS_DROP_RES, bias=bd, R=Rd)
    assert T.maxdiff(c, ref + bias.double() + R.double()) < TOL * scale
    c, _ = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_GELU_BWD, R=Rd)
    assert T.maxdiff(c, ref * gelu_grad(R.double())) < TOL * scale
    c, _ = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_ADD_RES, R=Rd)
    assert T.maxdiff(c, ref + R.double()) < TOL * scale
    c, _ = T.gemm(Ad, Bd, M, N, K, epi=_lib.EPI_BIAS_TANH, bias=bd)
    assert T.maxdiff(c, torch.tanh(ref + bias.double())) < TOL * scale


@pytest.mark.parametrize("M,N,K", [(200, 64, 96), (224, 256, 128)])
def test_gemm_dropout_epilogue_and_a_operand(M, N, K):
    rate, seed, step, sid = 0.2, 99, 7, 5
    A, B, bias, R = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.3), rnd(N, seed=3), rnd(M, N, seed=4)
    st = T.new_state(seed, step)
    keep = orc.dropout_keep_mask((M, N), rate, seed, step, sid).double()
    ref = (A.double() @ B.double() + bias.double()) * keep / (1 - rate) + R.double()
    c, _ = T.gemm(A.to(DEV), B.to(DEV), M, N, K, epi=_lib.EPI_BIAS_DROP_RES, bias=bias.to(DEV), R=R.to(DEV), rng=st,
                  drop_stream=sid, drop_rate=rate)
    assert T.maxdiff(c, ref) < 1e-4
    assert abs(float(keep.mean()) - 0.8) < 0.02
    # dropout applied to the A operand while it is loaded (backward of a dropped projection)
    keep_a = orc.dropout_keep_mask((M, K), rate, seed, step, sid).double()
    ref = (A.double() * keep_a / (1 - rate)) @ B.double()
    c, _ = T.gemm(A.to(DEV), B.to(DEV), M, N, K, rng=st, drop_stream=sid, drop_rate=rate, a_dropout=1)
    assert T.maxdiff(c, ref) < 1e-4


@pytest.mark.parametrize("M,K,rate", [(224, 64, 0.2), (96, 256, 0.0), (512, 256, 0.2), (32, 128, 0.1)])
def test_gemm_residual_layernorm_epilogue(M, K, rate):
    """B4R_EPI_BIAS_DROP_RES_LN (hidden size 64): z = R + dropout(A.B + bias) and LayerNorm(z) with its row statistics from
    one launch == the two-launch sequence's results (transformer block tails, Keras TransformerEncoderBlock)."""
    N, seed, step, sid, eps = 64, 31, 3, 9, 1e-12
    lib = _lib.load()
    A, B, bias, R = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.3), rnd(N, seed=3), rnd(M, N, seed=4)
    g, b = 1.0 + rnd(N, seed=5, scale=0.2), rnd(N, seed=6, scale=0.2)
    st = T.new_state(seed, step)
    keep = orc.dropout_keep_mask((M, N), rate, seed, step, sid).double() if rate > 0 else torch.ones(M, N, dtype=torch.float64)
    z_ref = (A.double() @ B.double() + bias.double()) * keep / (1 - rate) + R.double()
    mean_ref = z_ref.mean(1)
    var_ref = z_ref.var(1, unbiased=False)
    y_ref = (z_ref - mean_ref[:, None]) / torch.sqrt(var_ref[:, None] + eps) * g.double() + b.double()
    dev = [t.to(DEV) for t in (A, B, bias, R, g, b)]
    z = torch.full((M, N), float("nan"), device=DEV)
    y = torch.full((M, N), float("nan"), device=DEV)
    mean = torch.full((M,), float("nan"), device=DEV)
    rstd = torch.full((M,), float("nan"), device=DEV)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = T.P(dev[0]), K, T.P(dev[1]), N, T.P(z), N
    d.M, d.N, d.K, d.b_is_nk, d.epilogue = M, N, K, 0, _lib.EPI_BIAS_DROP_RES_LN
    d.bias, d.C2, d.ldc2, d.R, d.ldr, d.qscale = T.P(dev[2]), T.P(y), N, T.P(dev[3]), N, 1.0
    d.rng, d.drop_stream, d.drop_rate = (T.P(st) if rate > 0 else None), sid, rate
    d.ln_gamma, d.ln_beta, d.ln_mean, d.ln_rstd, d.ln_eps = T.P(dev[4]), T.P(dev[5]), T.P(mean), T.P(rstd), eps
    if lib.b4r_get_gemm_mode() != 1:   # exact-fp32 mode: no fused tail, and the request is refused
        assert lib.b4r_gemm_ln_supported(C.byref(d)) == 0
        assert lib.b4r_gemm_f32(C.byref(d), T.stream()) == -2   # B4R_E_SHAPE
        return
    assert lib.b4r_gemm_ln_supported(C.byref(d)) == 1
    _lib.check(lib.b4r_gemm_f32(C.byref(d), T.stream()), "b4r_gemm_f32")
    scale = float(z_ref.abs().max())                      # sums of K products of O(0.3): |z| reaches ~25 at K = 256
    assert T.maxdiff(z, z_ref) < 1e-5 * max(scale, 4.0)
    assert T.maxdiff(y, y_ref) < 5e-5                     # normalised values are O(1)
    assert T.maxdiff(mean, mean_ref) < 1e-5
    assert float(((rstd.cpu().double() * torch.sqrt(var_ref + eps)) - 1).abs().max()) < 1e-5
    # shapes the fused tail does not take are refused, not silently computed another way
    d.N = d.ldc = d.ldc2 = d.ldr = 128
    assert lib.b4r_gemm_ln_supported(C.byref(d)) == 0
    assert lib.b4r_gemm_f32(C.byref(d), T.stream()) == -2   # B4R_E_SHAPE


@pytest.mark.parametrize("M,K,gather", [(224, 64, False), (96, 128, False), (224, 64, True)])
def test_gemm_gelu_layernorm_epilogue(M, K, gather):
    """B4R_EPI_BIAS_GELU_LN (hidden size 64): the dense(gelu) -> LayerNorm transform of tfm MaskedLM from one launch,
    optionally on rows gathered like b4r_gather_rows does (P positions per sequence of length L, clamped into the sequence)."""
    N, eps = 64, 1e-12
    lib = _lib.load()
    A, B, bias = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.2), rnd(N, seed=3)
    src, idx, per, L = None, None, 7, 20
    if gather:
        src = rnd((M // per) * L, K, seed=11)
        idx = torch.randint(-1, L + 1, (M,), generator=torch.Generator().manual_seed(5))   # -1 and L are clamped
        A = src[idx.clamp(0, L - 1) + (torch.arange(M) // per) * L]
    g, b = 1.0 + rnd(N, seed=5, scale=0.2), rnd(N, seed=6, scale=0.2)
    pre_ref = A.double() @ B.double() + bias.double()
    u_ref = gelu(pre_ref)
    mean_ref, var_ref = u_ref.mean(1), u_ref.var(1, unbiased=False)
    t_ref = (u_ref - mean_ref[:, None]) / torch.sqrt(var_ref[:, None] + eps) * g.double() + b.double()
    dev = [t.to(DEV) for t in (A, B, bias, g, b)]
    u, t, pre = (torch.full((M, N), float("nan"), device=DEV) for _ in range(3))
    mean, rstd = torch.full((M,), float("nan"), device=DEV), torch.full((M,), float("nan"), device=DEV)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = T.P(dev[0]), K, T.P(dev[1]), N, T.P(u), N
    d.M, d.N, d.K, d.b_is_nk, d.epilogue, d.bias, d.qscale = M, N, K, 0, _lib.EPI_BIAS_GELU_LN, T.P(dev[2]), 1.0
    d.C2, d.ldc2, d.C3, d.ldc3 = T.P(t), N, T.P(pre), N
    d.ln_gamma, d.ln_beta, d.ln_mean, d.ln_rstd, d.ln_eps = T.P(dev[3]), T.P(dev[4]), T.P(mean), T.P(rstd), eps
    copy = torch.full((M, K), float("nan"), device=DEV)
    if gather:
        src_d, idx_d = src.to(DEV), idx.to(DEV)
        d.A, d.a_gather_idx, d.a_gather_add_per, d.a_gather_per, d.a_copy, d.a_copy_ld = T.P(src_d), T.P(idx_d), L, per, T.P(copy), K
    if lib.b4r_get_gemm_mode() != 1:
        assert lib.b4r_gemm_ln_supported(C.byref(d)) == 0
        assert lib.b4r_gemm_f32(C.byref(d), T.stream()) == -2   # B4R_E_SHAPE
        return
    assert lib.b4r_gemm_ln_supported(C.byref(d)) == 1
    _lib.check(lib.b4r_gemm_f32(C.byref(d), T.stream()), "b4r_gemm_f32")
    if gather:
        assert torch.equal(copy.cpu(), A)          # the gathered rows, bit for bit
        d.epilogue = _lib.EPI_BIAS_GELU            # gathering is only offered with this epilogue: refused elsewhere
        assert lib.b4r_gemm_f32(C.byref(d), T.stream()) == -1
    assert T.maxdiff(pre, pre_ref) < 5e-5
    assert T.maxdiff(u, u_ref) < 5e-5
    assert T.maxdiff(t, t_ref) < 2e-4      # the LayerNorm divides by the small spread of gelu outputs
    assert T.maxdiff(mean, mean_ref) < 1e-5


@pytest.mark.parametrize("M,K,embed", [(224, 192, False), (96, 256, False), (512, 64, False), (32, 128, False),
                                       (224, 192, True), (96, 64, True)])
def test_gemm_layernorm_backward_epilogue(M, K, embed):
    """B4R_EPI_ADD_RES_LN_BWD (hidden size 64): dz = LayerNorm'(A.B^T + R) with dgamma / dbeta from one launch == the
    input-gradient product followed by the stand-alone LayerNorm backward (torch autograd of the same normalisation)."""
    N, eps = 64, 1e-12
    lib = _lib.load()
    A, B, R = rnd(M, K, seed=1, scale=0.3), rnd(N, K, seed=2, scale=0.3), rnd(M, N, seed=4)
    z = rnd(M, N, seed=7, scale=2.0) + 0.5
    L, V, rate, seed, step, sid = 25, 50, 0.2, 11, 2, 0
    st = T.new_state(seed, step)
    if embed:   # the embedding stage: z = table[id] + pos[row % L] (ids outside [0, V) read row 0), dy through the dropout first
        table, pos = rnd(V, N, seed=8), rnd(L, N, seed=9)
        ids = torch.randint(-2, V + 2, (M,), generator=torch.Generator().manual_seed(3))
        safe = torch.where((ids < 0) | (ids >= V), torch.zeros_like(ids), ids)
        z = table[safe] + pos[torch.arange(M) % L]
    g = 1.0 + rnd(N, seed=5, scale=0.2)
    zd = z.double().requires_grad_(True)
    gd = g.double().requires_grad_(True)
    bd = torch.zeros(N, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.layer_norm(zd, (N,), gd, bd, eps)
    dy = A.double() @ B.double().t() + R.double()
    if embed:
        dy = dy * orc.dropout_keep_mask((M, N), rate, seed, step, sid).double() / (1 - rate)
    y.backward(dy)
    mean = z.double().mean(1)
    rstd = 1.0 / torch.sqrt(z.double().var(1, unbiased=False) + eps)
    dev = [t.to(DEV) for t in (A, B, R, z, g, mean.float(), rstd.float())]
    dz = torch.full((M, N), float("nan"), device=DEV)
    dgb = torch.full((2 * N,), float("nan"), device=DEV)
    scratch = torch.full((lib.b4r_gemm_ln_bwd_partial_floats(M),), float("nan"), device=DEV)
    d = _lib.GemmDesc()
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = T.P(dev[0]), K, T.P(dev[1]), K, T.P(dz), N
    d.M, d.N, d.K, d.b_is_nk, d.epilogue = M, N, K, 1, _lib.EPI_ADD_RES_LN_BWD
    d.R, d.ldr, d.qscale, d.C2 = T.P(dev[2]), N, 1.0, T.P(scratch)
    d.ln_z, d.ln_ldz, d.ln_gamma, d.ln_mean, d.ln_rstd = T.P(dev[3]), N, T.P(dev[4]), T.P(dev[5]), T.P(dev[6])
    d.ln_dgamma, d.ln_dbeta = dgb.data_ptr(), dgb.data_ptr() + 4 * N
    if embed:
        dev_e = [table.to(DEV), pos.to(DEV), ids.to(DEV)]
        d.ln_z, d.ln_table, d.ln_pos, d.ln_ids, d.ln_L, d.ln_V = None, T.P(dev_e[0]), T.P(dev_e[1]), T.P(dev_e[2]), L, V
        d.rng, d.drop_stream, d.drop_rate = T.P(st), sid, rate
    if lib.b4r_get_gemm_mode() != 1:
        assert lib.b4r_gemm_ln_supported(C.byref(d)) == 0
        assert lib.b4r_gemm_f32(C.byref(d), T.stream()) == -2   # B4R_E_SHAPE
        return
    assert lib.b4r_gemm_ln_supported(C.byref(d)) == 1
    _lib.check(lib.b4r_gemm_f32(C.byref(d), T.stream()), "b4r_gemm_f32")
    assert T.maxdiff(dz, zd.grad) < 5e-5 * max(1.0, float(zd.grad.abs().max()))
    scale = max(1.0, float(gd.grad.abs().max()), float(bd.grad.abs().max()))
    assert T.maxdiff(dgb[:N], gd.grad) < 2e-5 * scale * math.sqrt(M)
    assert T.maxdiff(dgb[N:], bd.grad) < 2e-5 * scale * math.sqrt(M)
    d.ln_dbeta = dgb.data_ptr()   # not the strip layout
    assert lib.b4r_gemm_f32(C.byref(d), T.stream()) == -1   # B4R_E_BADARG


@pytest.mark.parametrize("M,N,K,ldc", [(256, 3709, 64, 3712), (130, 37, 16, 64), (96, 64, 192, 64), (512, 64, 256, 64),
                                       (128, 128, 128, 128), (256, 384, 256, 384), (160, 1001, 128, 1004)])
def test_gemm_nt_vocab_projection(M, N, K, ldc):
    """C = A.B^T with B [N,K]: the tied projection T.E^T + b (tfm MaskedLM) incl. the padded leading dimension."""
    A, B, bias = rnd(M, K, seed=5), rnd(N, K, seed=6, scale=0.05), rnd(N, seed=7, scale=0.1)
    ref = A.double() @ B.double().t() + bias.double()
    c, _ = T.gemm(A.to(DEV), B.to(DEV), M, N, K, b_is_nk=1, epi=_lib.EPI_BIAS, bias=bias.to(DEV), ldc=ldc)
    assert T.maxdiff(c, ref) < TOL
    # with the pad columns of C declared scratch the register-operand bf16x3 kernel may take the shape (N % 4 != 0)
    c, _ = T.gemm(A.to(DEV), B.to(DEV), M, N, K, b_is_nk=1, epi=_lib.EPI_BIAS, bias=bias.to(DEV), ldc=ldc, c_pad_scratch=1)
    assert T.maxdiff(c, ref) < TOL


def test_gemm_k_tail_padded_rows():
    """dT = dlogits[M, V (ld Vp)] . E[V,H]: K = 3709 is neither a multiple of the K tile nor of 4."""
    M, V, Vp, H = 96, 3709, 3712, 64
    dl = torch.zeros(M, Vp)
    dl[:, :V] = rnd(M, V, seed=8, scale=0.02)
    dl[:, V:] = 1e30  # must never be read into the sum
    E = rnd(V, H, seed=9, scale=0.05)
    ref = dl[:, :V].double() @ E.double()
    c, _ = T.gemm(dl.to(DEV), E.to(DEV), M, H, V)
    assert T.maxdiff(c, ref) < TOL


# the last three: 128 x 128 output tiles (rx_gemm_tn128_kernel), incl. a row count that ends inside a chunk and a single tile
@pytest.mark.parametrize("R,Mo,No", [(1000, 64, 192), (517, 300, 64), (4096, 64, 64), (33, 5, 9), (3000, 256, 1024), (1237, 1024, 256),
                                     (70, 128, 128)])
def test_gemm_tn_weight_gradient(R, Mo, No):
    A, B = rnd(R, Mo, seed=10), rnd(R, No, seed=11)
    ref = A.double().t() @ B.double()
    out, cs, csa = T.gemm_tn(A.to(DEV), B.to(DEV), R, Mo, No, want_colsum=True, want_colsum_a=True)
    s = max(1.0, float(ref.abs().max()))
    assert T.maxdiff(out, ref) < 5e-5 * s
    assert T.maxdiff(cs, B.double().sum(0)) < 5e-5 * s
    assert T.maxdiff(csa, A.double().sum(0)) < 5e-5 * s
    # bitwise reproducible (ordered slab reduction, no atomics)
    out2, cs2, _ = T.gemm_tn(A.to(DEV), B.to(DEV), R, Mo, No, want_colsum=True)
    assert torch.equal(out, out2) and torch.equal(cs, cs2)


@pytest.mark.parametrize("R,Mo,rate", [(1000, 256, 0.2), (300, 128, 0.0)])
def test_gemm_tn_with_fused_input_gradient_and_gelu_tail(R, Mo, rate):
    """the FFN output layer's pair: dW2 = f^T.drop(dz) and dFpre = (drop(dz).W2^T) * gelu'(fpre) from one pass over dz."""
    No = 64
    seed, step, sid = 8, 3, 7
    lib = _lib.load()
    A, B, W, G = rnd(R, Mo, seed=1), rnd(R, No, seed=2), rnd(Mo, No, seed=3, scale=0.3), rnd(R, Mo, seed=4, scale=1.5)
    st = T.new_state(seed, step)
    keep = orc.dropout_keep_mask((R, No), rate, seed, step, sid).double() if rate > 0 else torch.ones(R, No, dtype=torch.float64)
    Bd = B.double() * keep / (1 - rate)
    dev = [t.to(DEV) for t in (A, B, W, G)]
    out = torch.full((Mo, No), float("nan"), device=DEV)
    cs = torch.full((No,), float("nan"), device=DEV)
    dx = torch.full((R, Mo), float("nan"), device=DEV)
    scratch = torch.empty(lib.b4r_gemm_tn_scratch_floats(R, Mo, No), device=DEV)
    d = _lib.GemmTnDesc()
    d.A, d.lda, d.B, d.ldb, d.out, d.ldo = T.P(dev[0]), Mo, T.P(dev[1]), No, T.P(out), No
    d.R, d.Mo, d.No, d.colsum = R, Mo, No, T.P(cs)
    d.rng, d.drop_stream, d.drop_rate, d.b_dropout = (T.P(st) if rate > 0 else None), sid, rate, 1
    d.dgrad_w, d.dgrad_ldw, d.dgrad_out, d.dgrad_ldo = T.P(dev[2]), No, T.P(dx), Mo
    d.dgrad_gelu_pre, d.dgrad_ldg = T.P(dev[3]), Mo
    if lib.b4r_get_gemm_mode() != 1:
        assert lib.b4r_gemm_tn_dgrad_supported(C.byref(d)) == 0
        return
    assert lib.b4r_gemm_tn_dgrad_supported(C.byref(d)) == 1
    _lib.check(lib.b4r_gemm_tn_f32(C.byref(d), T.P(scratch), T.stream()), "b4r_gemm_tn_f32")
    assert T.maxdiff(out, A.double().t() @ Bd) < 2e-5 * math.sqrt(R) * 4
    assert T.maxdiff(cs, Bd.sum(0)) < 2e-5 * math.sqrt(R) * 4
    assert T.maxdiff(dx, (Bd @ W.double().t()) * gelu_grad(G.double())) < 2e-4   # A&S erf in gelu' (1.5e-7) x |dx| up to ~10


@pytest.mark.parametrize("R,rate", [(1000, 0.2), (4096, 0.0), (200, 0.1), (70, 0.2)])
def test_gemm_tn_with_fused_input_gradient(R, rate):
    """b4r_gemm_tn_f32 with dgrad_out: dW = A^T.drop(B) (+ column sums) AND dX = drop(B).W^T from one pass over B (the pair
    of products behind y = x.W, hidden size 64) == the two separate products."""
    Mo = No = 64
    seed, step, sid = 5, 9, 6
    lib = _lib.load()
    A, B, W = rnd(R, Mo, seed=1), rnd(R, No, seed=2), rnd(Mo, No, seed=3, scale=0.3)
    st = T.new_state(seed, step)
    keep = orc.dropout_keep_mask((R, No), rate, seed, step, sid).double() if rate > 0 else torch.ones(R, No, dtype=torch.float64)
    Bd = B.double() * keep / (1 - rate)
    dev = [t.to(DEV) for t in (A, B, W)]
    out = torch.full((Mo, No), float("nan"), device=DEV)
    cs = torch.full((No,), float("nan"), device=DEV)
    dx = torch.full((R, Mo), float("nan"), device=DEV)
    scratch = torch.empty(lib.b4r_gemm_tn_scratch_floats(R, Mo, No), device=DEV)
    d = _lib.GemmTnDesc()
    d.A, d.lda, d.B, d.ldb, d.out, d.ldo = T.P(dev[0]), Mo, T.P(dev[1]), No, T.P(out), No
    d.R, d.Mo, d.No, d.colsum = R, Mo, No, T.P(cs)
    d.rng, d.drop_stream, d.drop_rate, d.b_dropout = (T.P(st) if rate > 0 else None), sid, rate, 1
    d.dgrad_w, d.dgrad_ldw, d.dgrad_out, d.dgrad_ldo = T.P(dev[2]), No, T.P(dx), Mo
    if lib.b4r_get_gemm_mode() != 1:
        assert lib.b4r_gemm_tn_dgrad_supported(C.byref(d)) == 0
        assert lib.b4r_gemm_tn_f32(C.byref(d), T.P(scratch), T.stream()) == -2   # B4R_E_SHAPE: refused, not computed otherwise
        return
    assert lib.b4r_gemm_tn_dgrad_supported(C.byref(d)) == 1
    _lib.check(lib.b4r_gemm_tn_f32(C.byref(d), T.P(scratch), T.stream()), "b4r_gemm_tn_f32")
    assert T.maxdiff(out, A.double().t() @ Bd) < 2e-5 * math.sqrt(R) * 4
    assert T.maxdiff(cs, Bd.sum(0)) < 2e-5 * math.sqrt(R) * 4
    assert T.maxdiff(dx, Bd @ W.double().t()) < 1e-4


@pytest.mark.parametrize("Mo,No", [(64, 64), (256, 128)])
def test_gemm_tn_dropout_on_b(Mo, No):
    R, rate, seed, step, sid = 700, 0.3, 5, 11, 2
    A, B = rnd(R, Mo, seed=12), rnd(R, No, seed=13)
    st = T.new_state(seed, step)
    keep = orc.dropout_keep_mask((R, No), rate, seed, step, sid).double()
    Bd = B.double() * keep / (1 - rate)
    out, cs, _ = T.gemm_tn(A.to(DEV), B.to(DEV), R, Mo, No, want_colsum=True, rng=st, drop_stream=sid, drop_rate=rate,
                           b_dropout=1)
    ref = A.double().t() @ Bd
    assert T.maxdiff(out, ref) < 1e-5 * float(ref.abs().max())
    assert T.maxdiff(cs, Bd.sum(0)) < 2e-4


@pytest.mark.parametrize("H", [32, 64, 128, 256, 512])
def test_layer_norm_fwd_bwd(H):
    rows = 333
    lib = _lib.load()
    z = rnd(rows, H, seed=14, scale=2.0) + 0.5
    gamma, beta, dy = rnd(H, seed=15) + 1.0, rnd(H, seed=16), rnd(rows, H, seed=17)
    zr = z.clone().double().requires_grad_(True)
    gr, br = gamma.clone().double().requires_grad_(True), beta.clone().double().requires_grad_(True)
    yr = orc.layer_norm(zr, gr, br, 1e-12)
    yr.backward(dy.double())
    zd, gd, bd = z.to(DEV), gamma.to(DEV), beta.to(DEV)
    y = torch.empty_like(zd)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    _lib.check(lib.b4r_ln_fwd(P(zd), rows, H, P(gd), P(bd), 1e-12, P(y), P(mean), P(rstd), stream()))
    assert T.maxdiff(y, yr) < 2e-5
    dz = torch.empty_like(zd)
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    scratch = torch.empty(lib.b4r_ln_bwd_scratch_floats(rows, H), device=DEV)
    dyd = dy.to(DEV)
    _lib.check(lib.b4r_ln_bwd(P(dyd), P(zd), P(mean), P(rstd), P(gd), rows, H, P(dz), P(dg), P(db), P(scratch), stream()))
    assert T.maxdiff(dz, zr.grad) < 5e-5
    assert T.maxdiff(dg, gr.grad) < 2e-4
    assert T.maxdiff(db, br.grad) < 2e-4


@pytest.mark.parametrize("rate", [0.0, 0.2])
def test_embedding_stage(rate):
    """x = dropout(LN(E[ids] + P[:L]))  bert4rec_encoder.py:198-211"""
    lib = _lib.load()
    B, L, V, H, Lmax, seed, step = 7, 23, 101, 64, 40, 77, 3
    ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(1))
    E, Pos = rnd(V, H, seed=18, scale=0.02), rnd(Lmax, H, seed=19, scale=0.02)
    gamma, beta = rnd(H, seed=20) + 1, rnd(H, seed=21)
    ref = orc.layer_norm((E[ids] + Pos[:L][None]).double(), gamma.double(), beta.double(), 1e-12)
    if rate > 0:
        ref = ref * orc.dropout_keep_mask((B, L, H), rate, seed, step, orc.STREAM_EMB).double() / (1 - rate)
    st = T.new_state(seed, step)
    out = torch.empty(B * L, H, device=DEV)
    mean, rstd = torch.empty(B * L, device=DEV), torch.empty(B * L, device=DEV)
    idd, Ed, Pd, gd, bd = ids.to(DEV), E.to(DEV), Pos.to(DEV), gamma.to(DEV), beta.to(DEV)  # keep alive: no temporaries
    _lib.check(lib.b4r_embed_ln_fwd(P(idd), B, L, P(Ed), V, P(Pd), P(gd), P(bd), H, 1e-12, P(out), P(mean), P(rstd), P(st),
                                    rate, stream()))
    assert T.maxdiff(out.view(B, L, H), ref) < 3e-4  # LN of ~0.03-magnitude rows: rstd ~ 35 amplifies fp32 rounding


def attention_reference(q, k, v, mask, rate=0.0, keep=None):
    """q,k,v [B,L,h,d] (q already scaled); mask [B,L] -> ctx [B,L,h,d], fp64"""
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) + (1.0 - mask.double())[:, None, None, :] * -1e9
    a = torch.softmax(s, dim=-1)
    if keep is not None:
        a = a * keep.double() / (1 - rate)
    return torch.einsum("bhqk,bkhd->bqhd", a, v)


@pytest.fixture(params=["tiles_by_length", "tiles32_always"])
def attention_tile_choice(request):
    """b4r_attn_bwd runs on 32-token tiles (one workgroup per sequence and head, b4r_attn32.hip) from L = 65 to 224, on round 1's
    16-token-tile kernels elsewhere; b4r_attn_fwd stays on 16-token tiles and writes the dropout decisions in both layouts.  The
    second parameter forces the 32-token tiles for every L <= 224, forward included."""
    lib = _lib.load()
    prev = lib.b4r_attn32_set_min_len(1 if request.param == "tiles32_always" else -1)
    prev_fwd = lib.b4r_attn32_set_core_fwd(1 if request.param == "tiles32_always" else -1)   # (default: only the backward uses them)
    yield request.param
    lib.b4r_attn32_set_min_len(prev)
    lib.b4r_attn32_set_core_fwd(prev_fwd)


@pytest.mark.parametrize("B,L,heads,rate", [(3, 50, 2, 0.0), (2, 200, 2, 0.0), (2, 200, 2, 0.2), (2, 64, 4, 0.1),
                                             (3, 17, 1, 0.0), (1, 130, 8, 0.0), (1, 256, 2, 0.0), (3, 65, 4, 0.3),
                                             (2, 224, 8, 0.1), (5, 100, 3, 0.5), (2, 33, 8, 0.2)])
def test_attention_fwd_bwd(B, L, heads, rate, attention_tile_choice):
    lib = _lib.load()
    H, d, seed, step, sid = heads * 32, 32, 21, 4, 9
    qkv = rnd(B * L, 3 * H, seed=22, scale=1.0)
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int64)
    if B > 1:
        mask[1, 0] = 0  # an interior hole: the mask is per key, not a length
    dctx = rnd(B * L, H, seed=23)
    x = qkv.double().view(B, L, 3, heads, d).clone().requires_grad_(True)
    keep = orc.dropout_keep_mask((B, heads, L, L), rate, seed, step, sid, orc.ATTN_PITCH) if rate > 0 else None
    ctx_ref = attention_reference(x[:, :, 0], x[:, :, 1], x[:, :, 2], mask, rate, keep)
    ctx_ref.backward(dctx.double().view(B, L, heads, d))
    st = T.new_state(seed, step)
    qd, md = qkv.to(DEV), mask.to(DEV)
    ctx = torch.full((B * L, H), float("nan"), device=DEV)
    lse = torch.empty(B * heads * L, device=DEV)
    bits = torch.zeros(lib.b4r_attn_keep_words(B, L, heads), dtype=torch.int32, device=DEV)
    x3 = lib.b4r_get_gemm_mode() == _lib.GEMM_BF16X3   # unit-variance inputs: scores of magnitude ~10, 2^-16 relative each
    _lib.check(lib.b4r_attn_fwd(P(qd), P(md), B, L, heads, P(ctx), P(lse), P(st), sid, rate, P(bits), stream()))
    assert T.maxdiff(ctx.view(B, L, heads, d), ctx_ref) < (3e-4 if x3 else 5e-5)
    dqkv = torch.full((B * L, 3 * H), float("nan"), device=DEV)
    qscale = 0.5
    dcd = dctx.to(DEV)
    _lib.check(lib.b4r_attn_bwd(P(qd), P(md), P(ctx), P(lse), P(dcd), B, L, heads, qscale, P(dqkv), P(st), sid, rate, P(bits), stream()))
    gref = x.grad.view(B * L, 3, H).clone()
    gref[:, 0] *= qscale
    # bf16x3: 2^-17 relative per product on scores of magnitude ~10 and gradients of magnitude up to ~10 (dropout 0.5 doubles them);
    # forward and backward may run on different tile shapes (their recomputed probabilities then differ by a few 1e-6 relative)
    assert T.maxdiff(dqkv.view(B * L, 3, H), gref) < (4e-4 * max(2.5, float(gref.abs().max())) if x3 else 2e-4)


@pytest.mark.parametrize("L,heads", [(20, 1), (100, 4)])
def test_attention_fully_masked_row_is_uniform(L, heads, attention_tile_choice):
    """Keras adds -1e9 (not -inf): a row whose keys are all masked attends uniformly (fp32 absorbs the scores), and its gradients are
    those of a softmax whose probabilities are all 1/L: ds = (dA - rowmean(dA)) / L still reaches q and k through the addition."""
    lib = _lib.load()
    B, H = 1, 32 * heads
    qkv = rnd(B * L, 3 * H, seed=30)
    mask = torch.zeros(B, L, dtype=torch.int64)
    ctx = torch.empty(B * L, H, device=DEV)
    lse = torch.empty(heads * L, device=DEV)
    qd, md = qkv.to(DEV), mask.to(DEV)
    _lib.check(lib.b4r_attn_fwd(P(qd), P(md), B, L, heads, P(ctx), P(lse), None, 0, 0.0, None, stream()))
    want = qkv[:, 2 * H:].double().mean(0, keepdim=True).expand(L, H)
    assert T.maxdiff(ctx, want) < 1e-5
    dctx = rnd(B * L, H, seed=31)
    dqkv = torch.full((B * L, 3 * H), float("nan"), device=DEV)
    dcd = dctx.to(DEV)
    _lib.check(lib.b4r_attn_bwd(P(qd), P(md), P(ctx), P(lse), P(dcd), B, L, heads, 0.5, P(dqkv), None, 0, 0.0, None, stream()))
    got = dqkv.cpu().double().view(L, 3, heads, 32)
    x = qkv.double().view(L, 3, heads, 32)
    do = dctx.double().view(L, heads, 32)
    dA = torch.einsum("qhd,khd->hqk", do, x[:, 2])
    ds = (dA - dA.mean(-1, keepdim=True)) / L
    assert T.maxdiff(got[:, 0], 0.5 * torch.einsum("hqk,khd->qhd", ds, x[:, 1])) < 2e-4
    assert T.maxdiff(got[:, 1], torch.einsum("hqk,qhd->khd", ds, x[:, 0])) < 2e-4
    assert T.maxdiff(got[:, 2], do.mean(0, keepdim=True).expand(L, heads, 32)) < 1e-5


def test_softmax_cross_entropy_and_metrics():
    """trainer_utils.py:12-23 (loss) and :49-60 / SparseCategoricalAccuracy (metrics)"""
    lib = _lib.load()
    M, V, ld = 123, 3709, 3712
    logits = rnd(M, V, seed=31, scale=2.0)
    y = torch.randint(0, V, (M,), generator=torch.Generator().manual_seed(5))
    y[::4] = 0  # ignored slots
    y[1] = int(logits[1].argmax())
    y[2] = int(logits[2].argmax())
    buf = torch.full((M, ld), 7.0)
    buf[:, :V] = logits
    bd = buf.to(DEV)
    st = T.new_state()
    rows = torch.empty(4 * M, device=DEV)
    _lib.check(lib.b4r_state_begin_step(P(st), stream()))
    yd = y.to(DEV)
    _lib.check(lib.b4r_softmax_ce(P(bd), M, V, ld, P(yd), P(rows), P(st), 1, stream()))
    f = T.state_floats(st)
    lr = logits.double().requires_grad_(True)
    valid = (y != 0)
    per = torch.logsumexp(lr, -1) - lr[torch.arange(M), y]
    loss_sum = (per * valid).sum()
    loss_sum.backward()
    assert abs(float(f[_lib.ST_LOSS_SUM]) - float(loss_sum)) < 1e-3 * max(1.0, float(loss_sum)) * 1e-2 + 1e-3
    assert float(f[_lib.ST_VALID]) == float(valid.sum())
    pred = logits.argmax(-1)
    assert float(f[_lib.ST_CORRECT_MASKED]) == float(((pred == y) & valid).sum())
    assert float(f[_lib.ST_CORRECT_ALL]) == float((pred == y).sum())
    assert float(f[_lib.ST_SLOTS_ALL]) == M
    out = bd.cpu()
    assert T.maxdiff(out[:, :V], lr.grad) < 2e-6
    assert float(out[:, V:].abs().max()) == 0.0


def test_adamw_matches_reference_update():
    """adam_w_optimizer.py:100-137: clip by global norm -> decoupled decay -> Keras Adam; schedule :22-36"""
    lib = _lib.load()
    n, n_decay = 4096 + 8, 3000
    hp_o = orc.AdamWConfig()
    hp = _lib.AdamWConfig(hp_o.init_lr, hp_o.end_lr, hp_o.num_train_steps, hp_o.num_warmup_steps, hp_o.weight_decay_rate,
                          hp_o.beta_1, hp_o.beta_2, hp_o.epsilon, hp_o.gradient_clip_norm)
    for step, gscale, count in [(0, 1.0, 1.0), (5, 40.0, 7.0), (150, 0.01, 3.0), (100, 1.0, 10.0)]:
        p0, g, m0, v0 = rnd(n, seed=40), rnd(n, seed=41, scale=gscale), rnd(n, seed=42, scale=0.01), rnd(n, seed=43, scale=0.01).abs()
        # reference: names chosen so that the first tensor decays and the second does not
        params = {"w/kernel": p0[:n_decay].clone(), "b/bias": p0[n_decay:].clone()}
        grads = {"w/kernel": g[:n_decay] / count, "b/bias": g[n_decay:] / count}
        m = {"w/kernel": m0[:n_decay].clone(), "b/bias": m0[n_decay:].clone()}
        v = {"w/kernel": v0[:n_decay].clone(), "b/bias": v0[n_decay:].clone()}
        gnorm = orc.adamw_apply(params, grads, m, v, step, hp_o)
        st = T.new_state(step=step)
        st.view(torch.float32)[_lib.ST_VALID] = count
        pd, gd, md, vd = p0.to(DEV), g.to(DEV), m0.to(DEV), v0.to(DEV)
        scratch = torch.empty(4096, device=DEV)
        _lib.check(lib.b4r_global_sqnorm(P(gd), n, P(scratch), P(st), stream()))
        _lib.check(lib.b4r_adamw_step(C.byref(hp), P(pd), P(gd), P(md), P(vd), n, n_decay, P(st), stream()))
        f = T.state_floats(st)
        assert abs(float(f[_lib.ST_GRAD_NORM]) - gnorm) < 1e-5 * max(1.0, gnorm)
        assert abs(float(f[_lib.ST_LR]) - float(orc.learning_rate(step, hp_o))) < 1e-12
        want_p = torch.cat([params["w/kernel"], params["b/bias"]])
        want_m = torch.cat([m["w/kernel"], m["b/bias"]])
        want_v = torch.cat([v["w/kernel"], v["b/bias"]])
        assert T.maxdiff(pd, want_p) < 2e-7 + 1e-6 * float(orc.learning_rate(step, hp_o)) / 1e-4
        assert T.maxdiff(md, want_m) < 1e-6 * max(1.0, gscale)
        assert T.maxdiff(vd, want_v) < 1e-6 * max(1.0, gscale * gscale)
        assert int(st.cpu()[_lib.ST_STEP:_lib.ST_STEP + 2].view(torch.int64)[0]) == step + 1


def _c_rank_oracle():
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    so = os.path.join(here, "librank_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", here])
    return C.CDLL(so)


@pytest.mark.parametrize("R,Cn,H,V", [(64, 101, 64, 3709), (5, 300, 256, 1000), (3, 3709, 64, 3709)])
def test_rank_candidates_bit_exact_against_c_oracle(R, Cn, H, V):
    """bert4rec_model.py:224-239 + bert4rec_evaluator.py:113-117; scores and ranked ids must be BIT-exact."""
    lib = _lib.load()
    co = _c_rank_oracle()
    g = torch.Generator().manual_seed(7)
    hidden, table, bias = rnd(R, H, seed=50), rnd(V, H, seed=51, scale=0.05), rnd(V, seed=52, scale=0.01)
    cand = torch.stack([torch.randperm(V, generator=g)[:Cn] for _ in range(R)]).to(torch.int64)
    # engineered ties: duplicate table rows so that equal scores occur; stable order must keep the lower index first
    table[cand[0, 5]] = table[cand[0, 9]]
    bias[cand[0, 5]] = bias[cand[0, 9]]
    gt = cand[:, -1].clone()
    hn, tn, bn, cn = hidden.numpy(), table.numpy(), bias.numpy(), cand.numpy()
    sc = np.zeros((R, Cn), np.float32)
    f32p, i64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    co.rank_oracle_scores(hn.ctypes.data_as(f32p), tn.ctypes.data_as(f32p), bn.ctypes.data_as(f32p),
                          cn.ctypes.data_as(i64p), C.c_int64(R), C.c_int64(Cn), C.c_int64(H), sc.ctypes.data_as(f32p))
    rk = np.zeros((R, Cn), np.int64)
    pos = np.zeros((R, Cn), np.int32)
    co.rank_oracle_rank(sc.ctypes.data_as(f32p), cn.ctypes.data_as(i64p), C.c_int64(R), C.c_int64(Cn),
                        rk.ctypes.data_as(i64p), pos.ctypes.data_as(i32p))
    ranking = torch.empty(R, Cn, dtype=torch.int64, device=DEV)
    gt_rank = torch.empty(R, dtype=torch.int32, device=DEV)
    scores = torch.empty(R, Cn, device=DEV)
    hd, td, bd, cd, gd = hidden.to(DEV), table.to(DEV), bias.to(DEV), cand.to(DEV), gt.to(DEV)  # keep alive
    _lib.check(lib.b4r_rank_candidates(P(hd), H, None, P(td), P(bd), H, table.shape[0], P(cd), R, Cn, P(gd), P(ranking), P(gt_rank),
                                       P(scores), None, 0, stream()))
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), sc.view(np.uint32)), "scores not bit-identical"
    assert np.array_equal(ranking.cpu().numpy(), rk)
    want_rank = orc.rank_of_ground_truth(rk, gt.numpy())
    assert np.array_equal(gt_rank.cpu().numpy().astype(np.int64), want_rank)
    # the numpy oracle agrees with the C oracle as well
    r2, _ = orc.rank_candidates(sc, cn)
    assert np.array_equal(r2, rk)
    assert sc[0, 5] == sc[0, 9] and pos[0, 5] + 1 == pos[0, 9]


@pytest.mark.parametrize("R,V,H,group_rows", [(3, 9000, 64, 0), (5, 26732, 64, 2), (2, 335423, 64, 0), (2, 10000, 256, 1)])
def test_whole_vocabulary_ranking_bit_exact_against_c_oracle(R, V, H, group_rows):
    """rank_items(items=None) (bert4rec_model.py:236): candidates = the whole vocabulary, no candidate list materialised;
    radix-argsort path of b4r_rank_candidates.  Beauty (54 545) / Reddit (335 423) sizes must work; ties keep the lower id."""
    lib = _lib.load()
    co = _c_rank_oracle()
    hidden, table, bias = rnd(R, H, seed=60), rnd(V, H, seed=61, scale=0.05), rnd(V, seed=62, scale=0.01)
    # engineered ties (equal table rows and biases), zeros of both signs and a block of identical items
    table[7] = table[3]; bias[7] = bias[3]
    table[V - 5:V] = table[11]; bias[V - 5:V] = bias[11]
    table[100] = 0.0; bias[100] = 0.0
    table[200] = 0.0; bias[200] = -0.0
    gt = torch.tensor([(37 * r + 5) % V for r in range(R)], dtype=torch.int64)
    hn, tn, bn = hidden.numpy(), table.numpy(), bias.numpy()
    cn = np.tile(np.arange(V, dtype=np.int64), (R, 1))
    sc = np.zeros((R, V), np.float32)
    f32p, i64p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    co.rank_oracle_scores(hn.ctypes.data_as(f32p), tn.ctypes.data_as(f32p), bn.ctypes.data_as(f32p),
                          cn.ctypes.data_as(i64p), C.c_int64(R), C.c_int64(V), C.c_int64(H), sc.ctypes.data_as(f32p))
    order = np.stack([np.argsort(-sc[r].astype(np.float64), kind="stable") for r in range(R)])   # = the counting rule
    ranking = torch.empty(R, V, dtype=torch.int64, device=DEV)
    gt_rank = torch.empty(R, dtype=torch.int32, device=DEV)
    scores = torch.empty(R, V, device=DEV)
    need = lib.b4r_rank_scratch_bytes(R, V)
    assert need == R * V * 20
    nbytes = need if group_rows == 0 else group_rows * V * 20   # a smaller scratch: rows are ranked in groups
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=DEV)
    hd, td, bd, gd = hidden.to(DEV), table.to(DEV), bias.to(DEV), gt.to(DEV)
    _lib.check(lib.b4r_rank_candidates(P(hd), H, None, P(td), P(bd), H, V, None, R, V, P(gd), P(ranking), P(gt_rank), P(scores),
                                       P(scratch), nbytes, stream()))
    torch.cuda.synchronize()
    assert np.array_equal(scores.cpu().numpy().view(np.uint32), sc.view(np.uint32)), "scores not bit-identical"
    assert np.array_equal(ranking.cpu().numpy(), order)
    want = np.array([1 + int(np.nonzero(order[r] == int(gt[r]))[0][0]) for r in range(R)])
    assert np.array_equal(gt_rank.cpu().numpy().astype(np.int64), want)
    pos3, pos7 = int(np.nonzero(order[0] == 3)[0][0]), int(np.nonzero(order[0] == 7)[0][0])
    assert sc[0, 3] == sc[0, 7] and pos3 + 1 == pos7
    # too little scratch is refused, never computed another way
    assert lib.b4r_rank_candidates(P(hd), H, None, P(td), P(bd), H, V, None, R, V, P(gd), P(ranking), P(gt_rank), None,
                                   P(scratch), V * 20 - 16, stream()) == -5


def test_rank_metric_sums_match_reference_known_answers():
    """b4r_rank_metrics against the reference's own known-answer values (tests/evaluators_tests/evaluation_metrics_tests.py:
    28-104, captured in tests/golden/reference_goldens.json) and against the C oracle's accumulation."""
    import json
    import os
    from bert4rec_amd import evaluation
    lib = _lib.load()
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_goldens.json")))
    for key, case in sorted(gold["evaluation_metrics"].items()):
        ms = evaluation.default_metrics()
        table = evaluation.gain_table(ms)
        fam = (C.c_int32 * len(table))(*[f for f, _ in table])
        cut = (C.c_int32 * len(table))(*[k for _, k in table])
        sums = torch.zeros(len(table), dtype=torch.float64, device=DEV)
        users = torch.zeros(1, dtype=torch.int64, device=DEV)
        ranks = torch.tensor(case["ranks"] + [0], dtype=torch.int32, device=DEV)   # a rank of 0 (ground truth absent) is skipped
        for _ in range(2):   # two batches accumulate
            _lib.check(lib.b4r_rank_metrics(P(ranks), ranks.numel(), fam, cut, len(table), P(sums), P(users), stream()))
        n = int(users.cpu()[0])
        assert n == 2 * len(case["ranks"])
        for m, g in zip(ms, sums.cpu().tolist()):
            m.absorb(g, n)
        got = {m.name: float(m.result()) for m in ms}
        for name, want in case["results"].items():
            if name == "Valid Ranks":
                assert got[name] == 2 * want
            else:
                assert abs(got[name] - want) < 1e-12, (key, name)


@pytest.mark.parametrize("M,V,H", [(100, 301, 64), (256, 3709, 64), (32, 33, 64), (1000, 64, 64), (5, 4000, 64),
                                   (96, 1000, 128), (200, 301, 256), (64, 5000, 256)])
def test_fused_mlm_head_matches_materialised_math(M, V, H):
    """b4r_mlm_head_fused_fwd / _bwd (train step without the [M,V] logits) against logits -> softmax CE -> autograd in
    fp64 (tfm MaskedLM head + trainer_utils.py:12-23,49-60): loss terms, metrics, dT, dE, dbias"""
    lib = _lib.load()
    T_, E_, b_ = rnd(M, H, seed=41, scale=1.5), rnd(V, H, seed=42, scale=0.3 * (64 / H) ** 0.5), rnd(V, seed=43, scale=0.5)
    g = torch.Generator().manual_seed(44)
    y = torch.randint(1, V, (M,), generator=g)
    y[::5] = 0                                           # ignored slots
    logits = T_.double() @ E_.double().T + b_.double()
    y[1] = int(logits[1].argmax())                       # a correct prediction
    if M > 7:
        y[7] = int(logits[7].argmax())
    Td, Ed, bd, yd = T_.to(DEV), E_.to(DEV), b_.to(DEV), y.to(DEV)
    scratch = torch.empty(lib.b4r_mlm_head_fused_scratch_floats(M, V, H), device=DEV)
    dT = torch.full((M, H), float("nan"), device=DEV)
    rows = torch.full((4 * M,), float("nan"), device=DEV)
    lse = torch.empty(M, device=DEV)
    lab = torch.empty(M, dtype=torch.int32, device=DEV)
    _lib.check(lib.b4r_mlm_head_fused_fwd(P(Td), P(Ed), P(bd), P(yd), M, V, H, P(scratch), P(dT), P(rows), P(lse), P(lab), 0, stream()))
    Tr = T_.double().requires_grad_(True)
    Er, br = E_.double().requires_grad_(True), b_.double().requires_grad_(True)
    lg = Tr @ Er.T + br
    valid = y != 0
    per = torch.logsumexp(lg, -1) - lg[torch.arange(M), y]
    (per * valid).sum().backward()
    r = rows.cpu().view(M, 4)
    assert T.maxdiff(r[:, 0], (per * valid).detach()) < 2e-4
    assert torch.equal(r[:, 1], valid.float())
    pred = logits.argmax(-1)
    assert torch.equal(r[:, 2], ((pred == y) & valid).float()) and torch.equal(r[:, 3], (pred == y).float())
    assert T.maxdiff(lse.cpu()[valid], torch.logsumexp(logits, -1)[valid]) < 2e-4 and bool(torch.isinf(lse.cpu()[~valid]).all())
    assert torch.equal(lab.cpu().long(), torch.where(valid, y, torch.full_like(y, -1)))
    assert T.maxdiff(dT, Tr.grad) < 1e-3 * float(Tr.grad.abs().max())
    dE = torch.full((V, H), float("nan"), device=DEV)
    db = torch.full((V,), float("nan"), device=DEV)
    _lib.check(lib.b4r_mlm_head_fused_bwd(P(Td), P(Ed), P(bd), P(lse), P(lab), M, V, H, P(scratch), P(dE), P(db), stream()))
    assert T.maxdiff(dE, Er.grad) < 1e-3 * float(Er.grad.abs().max())
    assert T.maxdiff(db, br.grad) < 1e-3 * float(br.grad.abs().max())


def test_sample_candidates_draws_like_numpy_choice_without_replacement():
    """b4r_sample_candidates vs the law of PopularRandomSampler.sample (popular_random_sampler.py:48-63: np.random.choice
    without replacement by popularity, excluded items dropped): ordered triples must follow the successive-draw
    probabilities p_a/S * p_b/(S-p_a) * p_c/(S-p_a-p_b) over the allowed items"""
    import itertools
    lib = _lib.load()
    V, C, R = 12, 3, 200000
    p = np.array([0.0, 0.0, 0.0, 0.20, 0.05, 0.15, 0.10, 0.02, 0.18, 0.0, 0.25, 0.05])   # specials and one unseen item: 0
    excl = [5, 11, -1, 40]                                      # two real exclusions, padding, out of range
    gt = 6
    with np.errstate(divide="ignore"):
        logp = torch.from_numpy(np.log(p).astype(np.float32)).to(DEV)
    ex = torch.tensor(excl, dtype=torch.int64).repeat(R, 1).to(DEV)
    gtd = torch.full((R,), gt, dtype=torch.int64, device=DEV)
    cand = torch.empty((R, C + 1), dtype=torch.int64, device=DEV)
    _lib.check(lib.b4r_sample_candidates(P(logp), V, P(ex), len(excl), P(gtd), R, C, 12345, P(cand), stream()))
    c = cand.cpu().numpy()
    assert (c[:, C] == gt).all()
    allowed = [v for v in range(V) if p[v] > 0 and v not in (5, 11, gt)]
    assert np.isin(c[:, :C], allowed).all()
    assert (c[:, 0] != c[:, 1]).all() and (c[:, 0] != c[:, 2]).all() and (c[:, 1] != c[:, 2]).all()
    S = sum(p[v] for v in allowed)
    code = c[:, 0] * V * V + c[:, 1] * V + c[:, 2]
    counts = np.bincount(code, minlength=V ** 3)
    worst = 0.0
    for a, b, d in itertools.permutations(allowed, 3):
        prob = p[a] / S * p[b] / (S - p[a]) * p[d] / (S - p[a] - p[b])
        exp = R * prob
        z = abs(counts[a * V * V + b * V + d] - exp) / np.sqrt(exp * (1 - prob) + 1.0)
        worst = max(worst, z)
    assert worst < 5.0, worst                                   # 336 cells, 5 sigma
    # determinism per seed, new draws per seed
    cand2 = torch.empty_like(cand)
    _lib.check(lib.b4r_sample_candidates(P(logp), V, P(ex), len(excl), P(gtd), R, C, 12345, P(cand2), stream()))
    assert torch.equal(cand, cand2)
    _lib.check(lib.b4r_sample_candidates(P(logp), V, P(ex), len(excl), P(gtd), R, C, 12346, P(cand2), stream()))
    assert not torch.equal(cand, cand2)
    # not enough drawable items: -1 entries (the Python layer raises like the reference)
    few = torch.empty((2, 8), dtype=torch.int64, device=DEV)
    _lib.check(lib.b4r_sample_candidates(P(logp), V, P(ex), len(excl), P(gtd), 2, 7, 1, P(few), stream()))
    assert int((few[:, :7] < 0).sum()) == 2 * (7 - len(allowed))


def test_mask_batch_follows_the_preprocessor_law():
    """b4r_mask_batch vs apply_dynamic_masking_task / process_element (dataloader_utils.py:186-261,
    bert4rec_preprocessor.py:48-116): per-row invariants for every length, the reference's count formula, uniform choice
    of positions and the [MASK] / random / unchanged proportions"""
    from bert4rec_amd.engine import Engine, make_model_config
    V, L, P = 500, 40, 12
    eng = Engine(make_model_config(V, 64, 1, 2, L, 64, 0.0, 0.0), "cuda")
    g = torch.Generator().manual_seed(0)
    B = 4000
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[:L] = torch.arange(1, L + 1)                                  # every length at least once
    toks = torch.randint(3, V, (B, L), generator=g)
    toks[torch.arange(L)[None, :] >= lens[:, None]] = 0
    out = {k: v.cpu() for k, v in eng.mask_batch(toks, P, selection_rate=0.2, seed=11).items()}
    assert torch.equal(out["labels"], toks)
    assert torch.equal(out["input_mask"], (toks != 0).long())
    for b in range(0, B, 7):
        n = int(lens[b])
        num = min(P, max(1, int(n * 0.2)))                             # dataloader_utils.py:214-215
        w = out["masked_lm_weights"][b]
        assert int(w.sum()) == num and bool((w[:num] == 1).all())
        pos = out["masked_lm_positions"][b, :num]
        assert bool((pos[1:] > pos[:-1]).all()) and int(pos.max()) < n   # ascending, inside the sequence
        assert torch.equal(out["masked_lm_ids"][b, :num], toks[b, pos])
        assert bool((out["masked_lm_positions"][b, num:] == 0).all()) and bool((out["masked_lm_ids"][b, num:] == 0).all())
        changed = out["input_word_ids"][b] != toks[b]
        assert int(changed.sum()) == num and bool((out["input_word_ids"][b, pos] == 1).all())   # always [MASK] by default
    # uniform choice: full-length rows, 8 of 40 positions each -> every position ~ 20 %
    full = lens == L
    hits = torch.zeros(L)
    for b in torch.nonzero(full).flatten().tolist():
        hits[out["masked_lm_positions"][b, :8]] += 1
    nfull = int(full.sum())
    assert float((hits / nfull - 0.2).abs().max()) < 5 * (0.2 * 0.8 / nfull) ** 0.5
    # 80 / 10 / 10 replacement
    out2 = {k: v.cpu() for k, v in eng.mask_batch(toks, P, 0.2, 0.8, 0.1, seed=12).items()}
    w2 = out2["masked_lm_weights"] == 1
    rows = torch.arange(B)[:, None].expand(B, P)[w2]
    posv = out2["masked_lm_positions"][w2]
    new, old = out2["input_word_ids"][rows, posv], toks[rows, posv]
    total = float(new.numel())
    frac_mask, frac_same = float((new == 1).sum()) / total, float((new == old).sum()) / total
    assert abs(frac_mask - 0.8) < 0.02 and abs(frac_same - 0.1) < 0.015 and bool(((new != 0) & (new != 2)).all())
    # another seed, another mask; same seed, same mask
    assert not torch.equal(out["masked_lm_positions"], eng.mask_batch(toks, P, 0.2, seed=13)["masked_lm_positions"].cpu())
    assert torch.equal(out["masked_lm_positions"], eng.mask_batch(toks, P, 0.2, seed=11)["masked_lm_positions"].cpu())
    # fine-tuning / evaluation rows: the last real token (mask_last_token_only)
    ft = {k: v.cpu() for k, v in eng.mask_batch(toks, P, finetune=True, seed=1).items()}
    assert torch.equal(ft["masked_lm_positions"][:, 0], lens - 1) and bool((ft["masked_lm_weights"].sum(1) == 1).all())
    assert torch.equal(ft["masked_lm_ids"][:, 0], toks[torch.arange(B), lens - 1])
    assert bool((ft["input_word_ids"][torch.arange(B), lens - 1] == 1).all())
    # the product of the kernel is a valid batch for the model
    cb, keep = eng.prepare_batch({k: v[:8] for k, v in out.items()})
    eng.init_parameters(seed=1)
    eng.forward(cb)
    torch.cuda.synchronize()


def test_fused_mlm_head_at_the_ml20m_vocabulary_matches_fp64_autograd():
    """b4r_mlm_head_fused_fwd / _bwd at V = 26 732, H = 256 (BASELINE.json configs[3]) against fp64 logits -> softmax CE -> autograd:
    loss rows, lse, dT, dE, d bias (M = 256 rows is enough to run every vocabulary tile of the NKH = 8 sweeps)."""
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    M, V, H = 256, 26732, 256
    Tm, E, bias = rnd(M, H, seed=80), rnd(V, H, seed=81, scale=0.03), rnd(V, seed=82, scale=0.01)
    y = torch.randint(1, V, (M,), generator=torch.Generator().manual_seed(83))
    y[::7] = 0                                                       # ignored slots (y_true == 0)
    T64, E64, b64 = Tm.double().requires_grad_(True), E.double().requires_grad_(True), bias.double().requires_grad_(True)
    logits = T64 @ E64.t() + b64
    lse_ref = torch.logsumexp(logits, 1)
    rows = lse_ref - logits[torch.arange(M), y]
    valid = (y != 0)
    (rows * valid).sum().backward()
    Td, Ed, bd, yd = Tm.to(DEV), E.to(DEV), bias.to(DEV), y.to(DEV)
    scratch = torch.empty(lib.b4r_mlm_head_fused_scratch_floats(M, V, H), dtype=torch.float32, device=DEV)
    dT = torch.empty(M, H, device=DEV); rs = torch.empty(4 * M, device=DEV); lse = torch.empty(M, device=DEV)
    lab = torch.empty(M, dtype=torch.int32, device=DEV); dE = torch.empty(V, H, device=DEV); db = torch.empty(V, device=DEV)
    _lib.check(lib.b4r_mlm_head_fused_fwd(P(Td), P(Ed), P(bd), P(yd), M, V, H, P(scratch), P(dT), P(rs), P(lse), P(lab), 0, stream()))
    _lib.check(lib.b4r_mlm_head_fused_bwd(P(Td), P(Ed), P(bd), P(lse), P(lab), M, V, H, P(scratch), P(dE), P(db), stream()))
    torch.cuda.synchronize()
    v = valid.to(DEV)
    assert T.maxdiff(lse[v], lse_ref.detach()[valid]) < 1e-4
    assert T.maxdiff(dT, T64.grad) < 1e-4
    assert T.maxdiff(dE, E64.grad) < 1e-4 * max(1.0, float(E64.grad.abs().max()))
    assert T.maxdiff(db, b64.grad) < 1e-4


@pytest.mark.parametrize("B,L,P", [(256, 200, 40), (7, 37, 5), (1, 16, 3), (256, 50, 20)])
def test_mlm_rows_lists_the_rows_the_head_gathers(B, L, P):
    """b4r_mlm_rows: one entry per slot, rows[m] = b*L + clamp(position), row_slot[m] = m for slots with an id, -1 for padded ones"""
    lib = _lib.load()
    batch = orc.synthetic_batch(B, L, P, 500, seed=B + L, ragged=True)
    pos, ids = batch["masked_lm_positions"].clone(), batch["masked_lm_ids"].clone()
    pos[0, -1], ids[0, -1] = L + 5, 0        # out of range on a padded slot: clamped like the gather
    want_rows = (torch.arange(B)[:, None] * L + pos.clamp(0, L - 1)).reshape(-1).tolist()
    want_slot = [m if int(v) != 0 else -1 for m, v in enumerate(ids.reshape(-1).tolist())]
    rows = torch.full((B * P,), -7, dtype=torch.int32, device=DEV)
    slot = torch.full((B * P,), -7, dtype=torch.int32, device=DEV)
    n = torch.zeros(1, dtype=torch.int32, device=DEV)
    pd, idd = pos.to(DEV), ids.to(DEV)
    _lib.check(lib.b4r_mlm_rows(T.P(pd), T.P(idd), B, L, P, T.P(rows), T.P(n), T.P(slot), stream()))
    torch.cuda.synchronize()
    assert int(n.cpu()[0]) == B * P
    assert rows.cpu().tolist() == want_rows and slot.cpu().tolist() == want_slot


def test_sample_candidates_takes_the_largest_gumbel_keys_in_order():
    """The sampler kernel's selection at an ML-1M-sized vocabulary (keys spread over all 256 threads of the workgroup: thread-local
    bests, wave argmax, cross-wave merge, owner rescan): with the keys restated on the host (key_v = log p_v - log(-log u_v), u_v from
    the counter hash of (seed, row, v)) the draws are the C largest keys among the allowed items, in descending order."""
    lib = _lib.load()
    V, C, R, E, seed = 3709, 100, 48, 200, (77 << 32) | 991
    g = torch.Generator().manual_seed(5)
    p = torch.rand(V, generator=g).double() ** 3
    p[:3] = 0.0
    p[torch.randint(3, V, (300,), generator=g)] = 0.0               # unseen items: never drawn
    logp32 = torch.log(p).float()
    ex = torch.randint(-1, V + 5, (R, E), generator=g)
    gt = torch.randint(3, V, (R,), generator=g)
    cand = torch.empty((R, C + 1), dtype=torch.int64, device=DEV)
    logpd, exd, gtd = logp32.to(DEV), ex.to(DEV), gt.to(DEV)
    _lib.check(lib.b4r_sample_candidates(P(logpd), V, P(exd), E, P(gtd), R, C, seed, P(cand), stream()))
    c = cand.cpu()
    assert torch.equal(c[:, C], gt)
    v = torch.arange(V, dtype=torch.int64)
    seed_lo, seed_hi = seed & 0xFFFFFFFF, seed >> 32
    for r in range(R):
        rk = orc._hash32_int((r * 0x9E3779B9 + seed_hi) & 0xFFFFFFFF)
        h = orc._hash32((orc._hash32(v ^ seed_lo) + rk) & 0xFFFFFFFF)
        u = ((h >> 9).double() + 0.5) / 8388608.0
        key = logp32.double() - torch.log(-torch.log(u))
        banned = set(int(x) for x in ex[r].tolist() if 0 <= x < V) | {int(gt[r])}
        key[list(banned)] = -float("inf")
        drawn = c[r, :C]
        assert len(set(drawn.tolist())) == C and not (set(drawn.tolist()) & banned) and bool((p[drawn] > 0).all())
        kd = key[drawn]
        assert bool((kd[:-1] >= kd[1:] - 1e-4).all()), r            # descending (the device uses fast logarithms: 1e-4 slack)
        rest = key.clone()
        rest[drawn] = -float("inf")
        assert float(kd.min()) >= float(rest.max()) - 1e-4, r      # nothing larger was left behind
