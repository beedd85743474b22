"""GPU parity tests of the fused encoder-layer halves (include/b4r.h: b4r_ffn_block_*, b4r_attn_block_*) through the C ABI
against an fp64 torch-CPU restatement of the same sub-graph of tfm TransformerEncoderBlock (post-LN), gradients from
torch.autograd.  Dropout masks are reproduced exactly with the oracle's restatement of the counter hash.
Tolerance: the bf16x3 split (3 bf16 products, fp32 accumulate) is ~2^-17 relative per product; 1e-4 absolute on O(1) values
(well inside the 1e-3 contract of BASELINE.json)."""
import ctypes as C
import math

import pytest
import torch

from bert4rec_amd import _lib
from oracle import bert4rec_oracle as orc
from tests import b4r_testlib as T
from tests.b4r_testlib import P, stream

pytestmark = [pytest.mark.gpu]
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.float32)


def gelu64(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def ln64(z, gamma, beta, eps):
    mean = z.mean(-1, keepdim=True)
    var = ((z - mean) ** 2).mean(-1, keepdim=True)
    return (z - mean) / torch.sqrt(var + eps) * gamma + beta, mean.squeeze(-1), (1.0 / torch.sqrt(var + eps)).squeeze(-1)


def ffn_inputs(N, seed):
    H, I = 64, 256
    t = dict(z1=rnd(N, H, seed=seed + 1), g1=1.0 + 0.2 * rnd(H, seed=seed + 2), be1=0.1 * rnd(H, seed=seed + 3),
             W1=rnd(H, I, seed=seed + 4, scale=0.15), b1=0.1 * rnd(I, seed=seed + 5), W2=rnd(I, H, seed=seed + 6, scale=0.1),
             b2=0.1 * rnd(H, seed=seed + 7), g2=1.0 + 0.2 * rnd(H, seed=seed + 8), be2=0.1 * rnd(H, seed=seed + 9),
             dx2=rnd(N, H, seed=seed + 10))
    return t


def ffn_reference(t, N, rate, seed, step, site, eps=1e-12):
    """fp64: x1 = LN1(z1); x2 = LN2(x1 + drop(gelu(x1 W1 + b1) W2 + b2)); loss = sum(x2 * dx2); autograd gradients."""
    d = {k: v.double().requires_grad_(k != "dx2") for k, v in t.items()}
    x1, mean1, rstd1 = ln64(d["z1"], d["g1"], d["be1"], eps)
    x1.retain_grad()
    y = gelu64(x1 @ d["W1"] + d["b1"]) @ d["W2"] + d["b2"]
    if rate > 0:
        keep = orc.dropout_keep_mask((N, 64), rate, seed, step, site).double()
        y = y * keep / (1.0 - rate)
    z2 = x1 + y
    z2.retain_grad()
    x2, mean2, rstd2 = ln64(z2, d["g2"], d["be2"], eps)
    (x2 * d["dx2"]).sum().backward()
    return dict(x1=x1.detach(), mean1=mean1.detach(), rstd1=rstd1.detach(), z2=z2.detach(), x2=x2.detach(),
                mean2=mean2.detach(), rstd2=rstd2.detach(), dz2=z2.grad, dz1=d["z1"].grad, dW1=d["W1"].grad, db1=d["b1"].grad,
                dW2=d["W2"].grad, db2=d["b2"].grad, dg1=d["g1"].grad, dbe1=d["be1"].grad)


@pytest.mark.parametrize("N,rate", [(16, 0.0), (200, 0.0), (1000, 0.2), (51200 // 8 + 7, 0.2), (31, 0.5)])
def test_ffn_block_matches_fp64_autograd(N, rate):
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_ffn_block_supported(64, 256) == 1
    seed, step, site = 4242, 3, 7
    t = ffn_inputs(N, seed=N)
    ref = ffn_reference(t, N, rate, seed, step, site)
    g = {k: v.to(DEV) for k, v in t.items()}
    x1 = ref["x1"].float().to(DEV)
    st = T.new_state(seed, step) if rate > 0 else None
    nan = float("nan")
    out = {k: torch.full(s, nan, dtype=torch.float32, device=DEV) for k, s in
           dict(z2=(N, 64), x2=(N, 64), mean2=(N,), rstd2=(N,), dz1=(N, 64), dW1=(64, 256), db1=(256,), dW2=(256, 64), db2=(64,),
                dln=(128,)).items()}
    d = _lib.FfnDesc()
    d.N, d.H, d.I = N, 64, 256
    d.x1, d.W1, d.b1, d.W2, d.b2 = P(x1), P(g["W1"]), P(g["b1"]), P(g["W2"]), P(g["b2"])
    d.ln_gamma, d.ln_beta, d.ln_eps = P(g["g2"]), P(g["be2"]), 1e-12
    d.rng, d.drop_stream, d.drop_rate = P(st), site, rate
    d.z2, d.x2, d.mean2, d.rstd2 = P(out["z2"]), P(out["x2"]), P(out["mean2"]), P(out["rstd2"])
    _lib.check(lib.b4r_ffn_block_fwd(C.byref(d), stream()), "b4r_ffn_block_fwd")
    torch.cuda.synchronize()
    for k in ("z2", "x2", "mean2"):
        assert T.maxdiff(out[k], ref[k]) < 1e-4, k
    assert T.maxdiff(out["rstd2"] / ref["rstd2"].float().to(DEV), torch.ones(N)) < 1e-4

    # backward: dz2 is what the output LayerNorm's backward hands over (taken from the reference graph)
    dz2 = ref["dz2"].float().to(DEV)
    mean1, rstd1 = ref["mean1"].float().to(DEV), ref["rstd1"].float().to(DEV)
    scratch = torch.empty(lib.b4r_ffn_block_bwd_scratch_floats(N), dtype=torch.float32, device=DEV)
    d.dz2, d.z1, d.mean1, d.rstd1, d.ln1_gamma = P(dz2), P(g["z1"]), P(mean1), P(rstd1), P(g["g1"])
    d.dz1, d.dW1, d.db1, d.dW2, d.db2, d.dln1_gamma = P(out["dz1"]), P(out["dW1"]), P(out["db1"]), P(out["dW2"]), P(out["db2"]), P(out["dln"])
    d.scratch = P(scratch)
    _lib.check(lib.b4r_ffn_block_bwd(C.byref(d), stream()), "b4r_ffn_block_bwd")
    torch.cuda.synchronize()
    assert T.maxdiff(out["dz1"], ref["dz1"]) < 1e-4

    def close(got, want, what):   # sums over N tokens: 2e-5 of the largest entry (they reach +-20 and more at N = 200)
        assert T.maxdiff(got, want) < 2e-5 * max(1.0, float(want.abs().max())), what
    for k in ("dW1", "db1", "dW2", "db2"):
        close(out[k], ref[k], k)
    close(out["dln"][:64], ref["dg1"], "dgamma1")
    close(out["dln"][64:], ref["dbe1"], "dbeta1")

    # bitwise reproducible: a second run of the backward gives identical bits (ordered partial sums, no atomics)
    first = {k: out[k].clone() for k in ("dz1", "dW1", "db1", "dW2", "db2", "dln")}
    _lib.check(lib.b4r_ffn_block_bwd(C.byref(d), stream()), "b4r_ffn_block_bwd")
    torch.cuda.synchronize()
    for k, v in first.items():
        assert torch.equal(v, out[k]), k


def test_ffn_block_refuses_other_shapes():
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_ffn_block_supported(128, 512) == 0
    d = _lib.FfnDesc()
    d.N, d.H, d.I = 16, 128, 512
    assert lib.b4r_ffn_block_fwd(C.byref(d), stream()) == -2   # B4R_E_SHAPE, never computed another way
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_F32))
    assert lib.b4r_ffn_block_supported(64, 256) == 0
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
