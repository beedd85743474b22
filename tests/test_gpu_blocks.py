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


@pytest.fixture(autouse=True, params=["tiles_by_length", "tiles32_always"])
def attention_tile_choice(request):
    """The attention block has two sets of kernels: 32-token tiles (b4r_attn32.hip), preferred from L = 65 on, and 16-token tiles
    (b4r_attn_block.hip) for shorter sequences (b4r_attn32_set_min_len).  Every test of this module runs under the default choice
    and with the 32-token tiles forced for every length they support, so both sets stay covered at L = 16, 37, 50."""
    lib = _lib.load()
    prev = lib.b4r_attn32_set_min_len(1 if request.param == "tiles32_always" else -1)
    yield request.param
    lib.b4r_attn32_set_min_len(prev)


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


# ---------------------------------------------------------------------------------------------------------------------------
# attention block
# ---------------------------------------------------------------------------------------------------------------------------
def attn_inputs(B, L, seed):
    H = 64
    t = dict(x=rnd(B * L, H, seed=seed + 1), Wqkv=rnd(H, 3 * H, seed=seed + 2, scale=0.1), bqkv=0.1 * rnd(3 * H, seed=seed + 3),
             Wo=rnd(H, H, seed=seed + 4, scale=0.15), bo=0.1 * rnd(H, seed=seed + 5), g1=1.0 + 0.2 * rnd(H, seed=seed + 6),
             be1=0.1 * rnd(H, seed=seed + 7))
    gen = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, L + 1, (B,), generator=gen)
    lens[0] = L
    if B > 2:
        lens[1] = 1
    mask = (torch.arange(L)[None, :] < lens[:, None]).to(torch.int64)
    if B > 3:
        mask[3] = 0          # a fully masked row: Keras' -1e9 gives a uniform softmax over ALL keys
    return t, mask


def attn_reference(t, mask, B, L, p_rate, o_rate, seed, step, site_p, site_o, eps=1e-12):
    """fp64 restatement of Keras MultiHeadAttention (2 heads of 32) + output dropout + residual + LayerNorm."""
    H, heads, dh = 64, 2, 32
    d = {k: v.double() for k, v in t.items()}
    x = d["x"].reshape(B, L, H)
    qkv = x @ d["Wqkv"] + d["bqkv"]
    q, k, v = (qkv[..., j * H:(j + 1) * H].reshape(B, L, heads, dh) for j in range(3))
    q = q * (1.0 / math.sqrt(dh))
    # the mask is added in fp32, as Keras does: -1e9 absorbs the scores of a fully masked row (ulp 64) -> uniform softmax
    adder = (1.0 - mask.float())[:, None, None, :] * torch.tensor(-1e9, dtype=torch.float32)
    s = (torch.einsum("bqhd,bkhd->bhqk", q, k).float() + adder).double()
    a = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if p_rate > 0:
        keep = orc.dropout_keep_mask((B, heads, L, L), p_rate, seed, step, site_p, row_pitch=orc.ATTN_PITCH).double()
        a = a * keep / (1.0 - p_rate)
    ctx = torch.einsum("bhqk,bkhd->bqhd", a, v).reshape(B * L, H)
    y = ctx @ d["Wo"] + d["bo"]
    if o_rate > 0:
        y = y * orc.dropout_keep_mask((B * L, H), o_rate, seed, step, site_o).double() / (1.0 - o_rate)
    z1 = d["x"] + y
    x1, mean1, rstd1 = ln64(z1, d["g1"], d["be1"], eps)
    qkv_scaled = torch.cat([q.reshape(B * L, H), k.reshape(B * L, H), v.reshape(B * L, H)], dim=1)
    return dict(qkv=qkv_scaled, ctx=ctx, lse=lse, z1=z1, x1=x1, mean1=mean1, rstd1=rstd1)


@pytest.mark.parametrize("B,L,p_rate,o_rate", [(3, 16, 0.0, 0.0), (5, 50, 0.0, 0.0), (4, 200, 0.2, 0.2), (6, 100, 0.0, 0.3),
                                              (2, 256, 0.2, 0.0), (7, 37, 0.5, 0.5), (3, 160, 0.2, 0.2), (2, 150, 0.0, 0.0)])
def test_attn_block_forward_matches_fp64(B, L, p_rate, o_rate):
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_attn_block_supported(64, 2, L) == 1
    seed, step, site_p, site_o = 991, 5, 1, 2
    t, mask = attn_inputs(B, L, seed=B * 1000 + L)
    ref = attn_reference(t, mask, B, L, p_rate, o_rate, seed, step, site_p, site_o)
    g = {k: v.to(DEV) for k, v in t.items()}
    st = T.new_state(seed, step) if (p_rate > 0 or o_rate > 0) else None
    N, nan = B * L, float("nan")
    out = {k: torch.full(s, nan, dtype=torch.float32, device=DEV) for k, s in
           dict(qkv=(N, 192), ctx=(N, 64), lse=(B, 2, L), z1=(N, 64), x1=(N, 64), mean1=(N,), rstd1=(N,)).items()}
    bits = torch.zeros(lib.b4r_attn_keep_words(B, L, 2), dtype=torch.int32, device=DEV)
    maskd = mask.to(DEV)
    d = _lib.AttnBlockDesc()
    d.B, d.L, d.H, d.heads = B, L, 64, 2
    d.x, d.input_mask = P(g["x"]), P(maskd)
    d.Wqkv, d.bqkv, d.Wo, d.bo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"]), P(g["bo"])
    d.ln_gamma, d.ln_beta, d.ln_eps = P(g["g1"]), P(g["be1"]), 1e-12
    d.rng, d.probs_stream, d.probs_rate, d.out_stream, d.out_rate = P(st), site_p, p_rate, site_o, o_rate
    d.qkv, d.ctx, d.lse, d.keep_bits = P(out["qkv"]), P(out["ctx"]), P(out["lse"]), P(bits)
    d.z1, d.x1, d.mean1, d.rstd1 = P(out["z1"]), P(out["x1"]), P(out["mean1"]), P(out["rstd1"])
    _lib.check(lib.b4r_attn_block_fwd(C.byref(d), stream()), "b4r_attn_block_fwd")
    torch.cuda.synchronize()
    for k in ("qkv", "ctx", "z1", "x1", "mean1"):   # scores reach +-5 here: 2e-4 on the softmax-weighted sums
        assert T.maxdiff(out[k], ref[k]) < 2e-4, k
    assert T.maxdiff(out["rstd1"] / ref["rstd1"].float().to(DEV), torch.ones(N)) < 1e-4
    # lse is stored relative to the sequence's largest mask adder (b4r_seq_amax): log(L) for a fully masked sequence, whose
    # log-sum-exp -1e9 + log(L) has no fp32 representation that keeps the log(L)
    lse_ref = ref["lse"].clone()
    dead = mask.sum(1) == 0
    lse_ref[dead] = math.log(L)
    assert float((out["lse"].cpu().double() - lse_ref).abs().max()) < 1e-4

    # the separate kernels of round 1 read what the block saved: b4r_attn_bwd on (qkv, ctx, lse, keep_bits) must agree with the
    # backward on the outputs of b4r_attn_fwd for the same inputs
    ctx2 = torch.empty(N, 64, device=DEV)
    lse2 = torch.empty(B, 2, L, device=DEV)
    bits2 = torch.zeros_like(bits)
    _lib.check(lib.b4r_attn_fwd(P(out["qkv"]), P(maskd), B, L, 2, P(ctx2), P(lse2), P(st), site_p, p_rate, P(bits2), stream()))
    torch.cuda.synchronize()
    assert T.maxdiff(ctx2, out["ctx"]) < 2e-5
    n_old = B * 2 * ((L + 15) // 16) * 128
    if p_rate > 0 and L > 224:   # the 16-token-tile block keeps round 1's layout of the decisions (first in the buffer); the
        assert torch.equal(bits2[:n_old], bits[:n_old])   # 32-token-tile block (L <= 224) one word per (query, 32-key tile) behind it
    if p_rate > 0 and L <= 224 and lib.b4r_attn32_set_min_len(-1) <= L:
        # the 32-token-tile forward ran: its decision words [b, head][key tile][query tile][slot], bit j = key 32 t + j, slot = the query's
        # place in the backward's register pairs (query 16s + 8a + 4h + b of the tile sits at 16s + 8a + 2b + h), must be
        # (a) the oracle's keep mask of the same (seed, step, site) bit for bit, (b) identical when the forward runs again
        NT = (L + 31) // 32
        words = bits[n_old:n_old + B * 2 * NT * NT * 32].cpu().view(B, 2, NT, NT, 32).to(torch.int64) & 0xFFFFFFFF
        keep = orc.dropout_keep_mask((B, 2, L, L), p_rate, seed, step, site_p, row_pitch=orc.ATTN_PITCH).bool()
        got = torch.zeros(B, 2, NT * 32, NT * 32, dtype=torch.bool)
        slot_of = [(r & 24) | ((r & 3) << 1) | ((r >> 2) & 1) for r in range(32)]
        words = words[..., slot_of]                                            # [B, 2, t, qt, query in tile]
        for j in range(32):   # got[b, h, query 32 qt + r, key 32 t + j]
            bit = ((words >> j) & 1).bool()
            got.view(B, 2, NT, 32, NT, 32)[:, :, :, :, :, j] = bit.permute(0, 1, 3, 4, 2)   # -> [B, 2, qt, r, t]
        live = mask.bool()
        for b in range(B):
            n_b = int(live[b].sum())
            if n_b == 0:
                continue
            assert torch.equal(got[b, :, :L, :L], keep[b]), f"sequence {b}: decision words differ from the oracle's keep mask"
        bits_again = torch.zeros_like(bits)
        d.keep_bits = P(bits_again)
        _lib.check(lib.b4r_attn_block_fwd(C.byref(d), stream()), "b4r_attn_block_fwd")
        torch.cuda.synchronize()
        assert torch.equal(bits_again, bits)


@pytest.mark.parametrize("B,L,Ps,p_rate,o_rate", [(4, 200, 40, 0.2, 0.2), (3, 160, 20, 0.0, 0.3), (2, 96, 33, 0.5, 0.0)])
def test_attn_block_forward_on_the_slots_rows_only(B, L, Ps, p_rate, o_rate):
    """out_slot_positions: only the named rows of the block's outputs are written, and only those queries are swept (compact query
    tiles, the softmax of a query merged from per-key-tile partials).  Against the dense launch on the same inputs: the slots' rows of
    ctx / z1 / x1 / mean1 / rstd1 / lse within the products' rounding, their decision words bit for bit, every other row untouched (NaN);
    padded slots repeat position 0; one sequence is fully masked."""
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    seed, step, site_p, site_o = 771, 9, 5, 6
    t, mask = attn_inputs(B, L, seed=B * 77 + L)
    mask[0, :] = 0
    g = {k: v.to(DEV) for k, v in t.items()}
    st = T.new_state(seed, step) if (p_rate > 0 or o_rate > 0) else None
    N, nan = B * L, float("nan")
    gen = torch.Generator().manual_seed(L + Ps)
    slot_pos = torch.stack([torch.randperm(L, generator=gen)[:Ps] for _ in range(B)]).to(torch.int64)
    slot_pos[:, Ps - 3:] = 0                                             # padded slots gather position 0 (tfm MaskedLM)
    slot_pos[1, 0] = 0                                                   # ... which a real slot may name too
    maskd, sposd = mask.to(DEV), slot_pos.to(DEV)
    res = []
    for slots in (False, True):
        out = {k: torch.full(sh, nan, dtype=torch.float32, device=DEV) for k, sh in
               dict(ctx=(N, 64), lse=(B, 2, L), z1=(N, 64), x1=(N, 64), mean1=(N,), rstd1=(N,)).items()}
        bits = torch.zeros(lib.b4r_attn_keep_words(B, L, 2), dtype=torch.int32, device=DEV)
        d = _lib.AttnBlockDesc()
        d.B, d.L, d.H, d.heads = B, L, 64, 2
        d.x, d.input_mask = P(g["x"]), P(maskd)
        d.Wqkv, d.bqkv, d.Wo, d.bo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"]), P(g["bo"])
        d.ln_gamma, d.ln_beta, d.ln_eps = P(g["g1"]), P(g["be1"]), 1e-12
        d.rng, d.probs_stream, d.probs_rate, d.out_stream, d.out_rate = P(st), site_p, p_rate, site_o, o_rate
        d.ctx, d.lse, d.keep_bits = P(out["ctx"]), P(out["lse"]), P(bits)
        d.z1, d.x1, d.mean1, d.rstd1 = P(out["z1"]), P(out["x1"]), P(out["mean1"]), P(out["rstd1"])
        if slots:
            d.out_slot_positions, d.out_slots = P(sposd), Ps
        _lib.check(lib.b4r_attn_block_fwd(C.byref(d), stream()), "b4r_attn_block_fwd")
        torch.cuda.synchronize()
        res.append((out, bits))
    (dense, bits_d), (comp, bits_c) = res
    rows = torch.unique((torch.arange(B)[:, None] * L + slot_pos).reshape(-1)).to(DEV)
    others = torch.ones(N, dtype=torch.bool, device=DEV)
    others[rows] = False
    for k in ("ctx", "z1", "x1"):
        assert T.maxdiff(comp[k][rows], dense[k][rows]) < 2e-5, k
        assert bool(torch.isnan(comp[k][others]).all()), k              # nothing else was written
    assert T.maxdiff(comp["mean1"][rows], dense["mean1"][rows]) < 2e-5
    assert T.maxdiff(comp["rstd1"][rows] / dense["rstd1"][rows], torch.ones(rows.numel())) < 2e-5
    lse_c, lse_d = comp["lse"].permute(0, 2, 1).reshape(N, 2), dense["lse"].permute(0, 2, 1).reshape(N, 2)
    assert T.maxdiff(lse_c[rows], lse_d[rows]) < 2e-5
    if p_rate > 0:   # the slots' decision words [b, head][key tile][query tile][slot of the query]: the dense launch's, bit for bit
        NT = (L + 31) // 32
        n_old = B * 2 * ((L + 15) // 16) * 128
        wd = bits_d[n_old:n_old + B * 2 * NT * NT * 32].view(B, 2, NT, NT, 32).cpu()
        wc = bits_c[n_old:n_old + B * 2 * NT * NT * 32].view(B, 2, NT, NT, 32).cpu()
        slot_of = [(r & 24) | ((r & 3) << 1) | ((r >> 2) & 1) for r in range(32)]
        for b in range(B):
            for q in set(int(v) for v in slot_pos[b]):
                assert torch.equal(wc[b, :, :, q // 32, slot_of[q % 32]], wd[b, :, :, q // 32, slot_of[q % 32]]), (b, q)


def attn_bwd_reference(t, mask, B, L, p_rate, o_rate, e_rate, seed, step, site_p, site_o, site_e, prev, eps=1e-12):
    """fp64 autograd through [previous LayerNorm (+ embedding dropout)] -> attention block; loss = sum(z1 * dz1)."""
    H, heads, dh = 64, 2, 32
    d = {k: v.double() for k, v in t.items()}
    g_prev, b_prev = prev["g"].double().requires_grad_(True), prev["b"].double().requires_grad_(True)
    zprev = prev["z"].double().requires_grad_(True)
    xn, mean_p, rstd_p = ln64(zprev, g_prev, b_prev, eps)
    x = xn
    if e_rate > 0:
        x = x * orc.dropout_keep_mask((B * L, H), e_rate, seed, step, site_e).double() / (1.0 - e_rate)
    x3 = x.reshape(B, L, H)
    qkv = x3 @ d["Wqkv"] + d["bqkv"]
    qkv.retain_grad()
    q, k, v = (qkv[..., j * H:(j + 1) * H].reshape(B, L, heads, dh) for j in range(3))
    q = q * (1.0 / math.sqrt(dh))
    adder = ((1.0 - mask.float())[:, None, None, :] * torch.tensor(-1e9, dtype=torch.float32)).double()
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) + adder
    # a fully masked row: the fp32 addition of -1e9 rounds every score away and the softmax is exactly uniform, while the gradient
    # of that addition is still the identity (autograd / tf.gradients do not differentiate the rounding): value 0, gradient 1
    dead = (mask.sum(1) == 0)
    if bool(dead.any()):
        s = torch.where(dead[:, None, None, None], s - s.detach(), s)
    a = torch.softmax(s, dim=-1)
    if p_rate > 0:
        a = a * orc.dropout_keep_mask((B, heads, L, L), p_rate, seed, step, site_p, row_pitch=orc.ATTN_PITCH).double() / (1.0 - p_rate)
    ctx = torch.einsum("bhqk,bkhd->bqhd", a, v).reshape(B * L, H)
    y = ctx @ d["Wo"] + d["bo"]
    if o_rate > 0:
        y = y * orc.dropout_keep_mask((B * L, H), o_rate, seed, step, site_o).double() / (1.0 - o_rate)
    z1 = x + y
    (z1 * t["dz1"].double()).sum().backward()
    return dict(x=x.detach(), mean_p=mean_p.detach(), rstd_p=rstd_p.detach(), dqkv=qkv.grad.reshape(B * L, 3 * H),
                dzprev=zprev.grad, dg=g_prev.grad, db=b_prev.grad)


@pytest.mark.parametrize("B,L,p_rate,o_rate,embed", [(3, 16, 0.0, 0.0, False), (5, 50, 0.0, 0.0, False), (4, 200, 0.2, 0.2, False),
                                                     (6, 100, 0.0, 0.3, True), (3, 208, 0.2, 0.0, False), (7, 37, 0.5, 0.5, True),
                                                     (3, 160, 0.2, 0.2, False), (2, 112, 0.0, 0.0, True)])
def test_attn_block_backward_matches_fp64_autograd(B, L, p_rate, o_rate, embed, attention_tile_choice):
    """b4r_attn_block_bwd (q, k, v recomputed, every score block formed once, dK / dV accumulated in LDS) against autograd, on
    what b4r_attn_block_fwd saved (ctx, lse, keep_bits); both LayerNorm variants in front of the block."""
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_attn_block_bwd_supported(64, 2, L) == 1 and lib.b4r_attn_block_bwd_supported(64, 2, 209) == 0
    seed, step, site_p, site_o, site_e = 313, 9, 1, 2, 0
    N, V = B * L, 97
    t, mask = attn_inputs(B, L, seed=B * 1000 + L + 7)
    t["dz1"] = rnd(N, 64, seed=B + L)
    e_rate = 0.25 if embed else 0.0
    if embed:
        ids = torch.randint(0, V, (B, L), generator=torch.Generator().manual_seed(5))
        table, pos = rnd(V, 64, seed=70), 0.3 * rnd(L, 64, seed=71)
        zprev = table[ids.reshape(-1)] + pos.repeat(B, 1)
    else:
        zprev = rnd(N, 64, seed=72)
    prev = dict(z=zprev, g=1.0 + 0.2 * rnd(64, seed=73), b=0.1 * rnd(64, seed=74))
    ref = attn_bwd_reference(t, mask, B, L, p_rate, o_rate, e_rate, seed, step, site_p, site_o, site_e, prev)
    g = {k: v.to(DEV) for k, v in t.items()}
    st = T.new_state(seed, step) if (p_rate > 0 or o_rate > 0 or e_rate > 0) else None
    nan = float("nan")
    x = ref["x"].float().to(DEV)
    maskd = mask.to(DEV)
    # forward first (what the backward consumes: ctx, lse, keep_bits); qkv is NOT stored
    out = {k: torch.full(s_, nan, dtype=torch.float32, device=DEV) for k, s_ in
           dict(ctx=(N, 64), lse=(B, 2, L), z1=(N, 64), x1=(N, 64), dqkv=(N, 192), da=(N, 64), dln=(128,)).items()}
    bits = torch.zeros(lib.b4r_attn_keep_words(B, L, 2), dtype=torch.int32, device=DEV)
    fd = _lib.AttnBlockDesc()
    fd.B, fd.L, fd.H, fd.heads = B, L, 64, 2
    fd.x, fd.input_mask = P(x), P(maskd)
    fd.Wqkv, fd.bqkv, fd.Wo, fd.bo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"]), P(g["bo"])
    fd.ln_gamma, fd.ln_beta, fd.ln_eps = P(g["g1"]), P(g["be1"]), 1e-12
    fd.rng, fd.probs_stream, fd.probs_rate, fd.out_stream, fd.out_rate = P(st), site_p, p_rate, site_o, o_rate
    fd.qkv, fd.ctx, fd.lse, fd.keep_bits = None, P(out["ctx"]), P(out["lse"]), P(bits)
    fd.z1, fd.x1 = P(out["z1"]), P(out["x1"])
    _lib.check(lib.b4r_attn_block_fwd(C.byref(fd), stream()), "b4r_attn_block_fwd")
    mean_p, rstd_p = ref["mean_p"].float().to(DEV), ref["rstd_p"].float().to(DEV)
    zp, gp = prev["z"].to(DEV), prev["g"].to(DEV)
    scratch = torch.empty(lib.b4r_attn_block_bwd_scratch_floats(B), dtype=torch.float32, device=DEV)
    bd = _lib.AttnBlockBwdDesc()
    bd.B, bd.L, bd.H, bd.heads = B, L, 64, 2
    bd.x, bd.dz1, bd.ctx, bd.lse, bd.keep_bits, bd.input_mask = P(x), P(g["dz1"]), P(out["ctx"]), P(out["lse"]), P(bits), P(maskd)
    bd.Wqkv, bd.bqkv, bd.Wo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"])
    bd.rng, bd.probs_stream, bd.probs_rate, bd.out_stream, bd.out_rate = P(st), site_p, p_rate, site_o, o_rate
    bd.prev_mean, bd.prev_rstd, bd.prev_gamma = P(mean_p), P(rstd_p), P(gp)
    if embed:
        idd, tabd, posd = ids.to(DEV), table.to(DEV), pos.to(DEV)
        bd.emb_ids, bd.emb_table, bd.emb_pos, bd.emb_vocab, bd.emb_stream, bd.emb_rate = P(idd), P(tabd), P(posd), V, site_e, e_rate
    else:
        bd.prev_z = P(zp)
    bd.dqkv, bd.dx_prev, bd.dprev_gamma, bd.scratch = P(out["dqkv"]), P(out["da"]), P(out["dln"]), P(scratch)
    _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()), "b4r_attn_block_bwd")
    torch.cuda.synchronize()

    def close(got, want, what, rel=3e-5):
        err = (got.detach().cpu().double() - want).abs()
        assert float(err.max()) < rel * max(1.0, float(want.abs().max())) + 1e-5, (what, float(err.max()), int(err.argmax()))
    close(out["dqkv"], ref["dqkv"], "dqkv")
    close(out["da"], ref["dzprev"], "dx_prev")
    close(out["dln"][:64], ref["dg"], "dgamma")
    close(out["dln"][64:], ref["db"], "dbeta")
    first = {k: out[k].clone() for k in ("dqkv", "da", "dln")}
    _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()), "b4r_attn_block_bwd")
    torch.cuda.synchronize()
    for k, v in first.items():
        assert torch.equal(v, out[k]), k     # ordered accumulation of dK / dV in LDS: bitwise reproducible

    # the same launch can form dWqkv = x^T.dqkv and dbqkv itself (no [N, 3H] round trip): ordered sums over per-sequence partials
    dw = torch.full((64, 192), nan, dtype=torch.float32, device=DEV)
    dbq = torch.full((192,), nan, dtype=torch.float32, device=DEV)
    dws = torch.empty(lib.b4r_attn_block_bwd_dw_scratch_floats(B), dtype=torch.float32, device=DEV)
    out["da"].fill_(nan)
    dwo = torch.full((64, 64), nan, dtype=torch.float32, device=DEV)
    dbo = torch.full((64,), nan, dtype=torch.float32, device=DEV)
    bd.dqkv, bd.dWqkv, bd.dbqkv, bd.dw_scratch, bd.dWo, bd.dbo = None, P(dw), P(dbq), P(dws), P(dwo), P(dbo)
    _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()), "b4r_attn_block_bwd (weight gradients inside)")
    torch.cuda.synchronize()
    close(dw, ref["x"].T @ ref["dqkv"], "dWqkv")
    close(dbq, ref["dqkv"].sum(0), "dbqkv")
    dy = t["dz1"].double()
    if o_rate > 0:
        dy = dy * orc.dropout_keep_mask((N, 64), o_rate, seed, step, site_o).double() / (1.0 - o_rate)
    close(dwo, out["ctx"].detach().cpu().double().T @ dy, "dWo")
    close(dbo, dy.sum(0), "dbo")
    if attention_tile_choice == "tiles32_always" or L >= 65:
        assert torch.equal(out["da"], first["da"])
    else:   # a short sequence under the default choice: the first call ran on 16-token tiles, this one (weight gradients inside) on 32
        close(out["da"], ref["dzprev"], "dx_prev (32-token tiles)")
    keep = [v.clone() for v in (dw, dbq, dwo, dbo)]
    _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()), "b4r_attn_block_bwd (weight gradients inside)")
    torch.cuda.synchronize()
    assert all(torch.equal(a, b_) for a, b_ in zip(keep, (dw, dbq, dwo, dbo)))

    # sparse dz1 (the last layer of a train step): only the rows named by (slot positions, slot ids != 0) exist; the launch must give
    # what it gives for the dense tensor with zeros elsewhere, without ever reading the other rows (NaN there).  With at most 64 slots
    # on three or more token tiles the sweep walks the slots as its only queries (compact query tiles, another summation order):
    # equal within the split products' rounding there, bit for bit otherwise -- and bit for bit between two runs of itself
    Ps = 5
    gen = torch.Generator().manual_seed(B + L)
    slot_pos = torch.stack([torch.randperm(L, generator=gen)[:Ps] if L >= Ps else torch.arange(Ps) % L for _ in range(B)]).to(torch.int64)
    slot_id = torch.randint(3, 90, (B, Ps), generator=gen).to(torch.int64)
    slot_id[:, -1] = 0                                                   # a padded slot: its row carries nothing
    rows = (torch.arange(B)[:, None] * L + slot_pos)[slot_id != 0]
    dense = torch.zeros(N, 64)
    dense[rows] = t["dz1"][rows]
    sparse = torch.full((N, 64), nan)
    sparse[rows] = t["dz1"][rows]
    res = []
    for dz, slots in ((dense, False), (sparse, True)):
        dzd, sposd, sidd = dz.to(DEV), slot_pos.to(DEV), slot_id.to(DEV)
        bd.dz1 = P(dzd)
        bd.dz1_slot_positions, bd.dz1_slot_ids, bd.dz1_slots = (P(sposd), P(sidd), Ps) if slots else (None, None, 0)
        for v in (dw, dbq, dwo, dbo, out["da"], out["dln"]):
            v.fill_(nan)
        _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()), "b4r_attn_block_bwd (sparse dz1)")
        torch.cuda.synchronize()
        res.append([v.clone() for v in (dw, dbq, dwo, dbo, out["da"], out["dln"])])
    assert all(bool(torch.isfinite(v).all()) for v in res[1])
    compact_queries = L > 64   # (three or more 32-token tiles: the 32-token-tile kernel under either tile choice)
    for name, a, b_ in zip(("dWqkv", "dbqkv", "dWo", "dbo", "dx_prev", "dprev_gamma"), *res):
        if compact_queries:
            assert float((a - b_).abs().max()) < 1e-4 * max(1.0, float(a.abs().max())), (name, float((a - b_).abs().max()))
        else:
            assert torch.equal(a, b_), (name, float((a - b_).abs().max()))
    for v in (dw, dbq, dwo, dbo, out["da"], out["dln"]):
        v.fill_(nan)
    _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()), "b4r_attn_block_bwd (sparse dz1, again)")
    torch.cuda.synchronize()
    for name, a, b_ in zip(("dWqkv", "dbqkv", "dWo", "dbo", "dx_prev", "dprev_gamma"), res[1], (dw, dbq, dwo, dbo, out["da"], out["dln"])):
        assert torch.equal(a, b_), name


# ---------------------------------------------------------------------------------------------------------------------------
# one whole encoder layer = the two halves (b4r_encoder_layer_fwd / _bwd)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,L,rate", [(4, 200, 0.2), (3, 37, 0.0)])
def test_encoder_layer_is_the_two_halves_plus_the_two_weight_gradient_products(B, L, rate):
    """bert4rec_encoder.py:220-222 (one TransformerEncoderBlock call): the layer entry points give bit for bit what the halves give
    when called one by one (each half is checked against fp64 autograd above), and the two products they add are
    dWo = ctx^T . dropmask(dz1), dbo = its column sums, dWqkv = x^T . dqkv, dbqkv = column sums of dqkv."""
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_encoder_layer_supported(64, 2, 256, L) == 1 and lib.b4r_encoder_layer_supported(64, 2, 512, L) == 0
    seed, step, N = 77, 2, B * L
    t, mask = attn_inputs(B, L, seed=L)
    f = ffn_inputs(N, seed=L + 1)
    g = {k: v.to(DEV) for k, v in {**t, **{k: f[k] for k in ("W1", "b1", "W2", "b2", "g2", "be2")}}.items()}
    maskd = mask.to(DEV)
    st = T.new_state(seed, step) if rate > 0 else None
    dz2 = rnd(N, 64, seed=5).to(DEV)
    zprev, gprev = rnd(N, 64, seed=6).to(DEV), (1.0 + 0.2 * rnd(64, seed=7)).to(DEV)
    mean_p = zprev.mean(1).contiguous()
    rstd_p = (zprev.var(1, unbiased=False) + 1e-12).rsqrt().contiguous()
    bits = torch.zeros(lib.b4r_attn_keep_words(B, L, 2), dtype=torch.int32, device=DEV)
    shapes = dict(ctx=(N, 64), lse=(B, 2, L), z1=(N, 64), x1=(N, 64), mean1=(N,), rstd1=(N,), z2=(N, 64), x2=(N, 64), mean2=(N,),
                  rstd2=(N,), dz1=(N, 64), dW1=(64, 256), db1=(256,), dW2=(256, 64), db2=(64,), dln1=(128,), dqkv=(N, 192),
                  da=(N, 64), dlnp=(128,), dWo=(64, 64), dbo=(64,), dWqkv=(64, 192), dbqkv=(192,))

    def run(layer_calls, store_x1=True):
        o = {k: torch.full(s_, float("nan"), dtype=torch.float32, device=DEV) for k, s_ in shapes.items()}
        bits.zero_()
        ad = _lib.AttnBlockDesc()
        ad.B, ad.L, ad.H, ad.heads = B, L, 64, 2
        ad.x, ad.input_mask = P(g["x"]), P(maskd)
        ad.Wqkv, ad.bqkv, ad.Wo, ad.bo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"]), P(g["bo"])
        ad.ln_gamma, ad.ln_beta, ad.ln_eps = P(g["g1"]), P(g["be1"]), 1e-12
        ad.rng, ad.probs_stream, ad.probs_rate, ad.out_stream, ad.out_rate = P(st), 1, rate, 2, rate
        ad.ctx, ad.lse, ad.keep_bits = P(o["ctx"]), P(o["lse"]), P(bits)
        ad.z1, ad.x1, ad.mean1, ad.rstd1 = P(o["z1"]), (P(o["x1"]) if store_x1 else None), P(o["mean1"]), P(o["rstd1"])
        fd = _lib.FfnDesc()
        fd.N, fd.H, fd.I = N, 64, 256
        fd.x1, fd.W1, fd.b1, fd.W2, fd.b2 = (P(o["x1"]) if store_x1 else None), P(g["W1"]), P(g["b1"]), P(g["W2"]), P(g["b2"])
        fd.ln1_beta = P(g["be1"])
        fd.ln_gamma, fd.ln_beta, fd.ln_eps = P(g["g2"]), P(g["be2"]), 1e-12
        fd.rng, fd.drop_stream, fd.drop_rate = P(st), 3, rate
        fd.z2, fd.x2, fd.mean2, fd.rstd2 = P(o["z2"]), P(o["x2"]), P(o["mean2"]), P(o["rstd2"])
        fd.dz2, fd.z1, fd.mean1, fd.rstd1, fd.ln1_gamma = P(dz2), P(o["z1"]), P(o["mean1"]), P(o["rstd1"]), P(g["g1"])
        fd.dz1, fd.dW1, fd.db1, fd.dW2, fd.db2, fd.dln1_gamma = P(o["dz1"]), P(o["dW1"]), P(o["db1"]), P(o["dW2"]), P(o["db2"]), P(o["dln1"])
        scr_f = torch.empty(lib.b4r_ffn_block_bwd_scratch_floats(N), dtype=torch.float32, device=DEV)
        fd.scratch = P(scr_f)
        bd = _lib.AttnBlockBwdDesc()
        bd.B, bd.L, bd.H, bd.heads = B, L, 64, 2
        bd.x, bd.dz1, bd.ctx, bd.lse, bd.keep_bits, bd.input_mask = P(g["x"]), P(o["dz1"]), P(o["ctx"]), P(o["lse"]), P(bits), P(maskd)
        bd.Wqkv, bd.bqkv, bd.Wo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"])
        bd.rng, bd.probs_stream, bd.probs_rate, bd.out_stream, bd.out_rate = P(st), 1, rate, 2, rate
        bd.prev_z, bd.prev_mean, bd.prev_rstd, bd.prev_gamma = P(zprev), P(mean_p), P(rstd_p), P(gprev)
        scr_a = torch.empty(lib.b4r_attn_block_bwd_scratch_floats(B), dtype=torch.float32, device=DEV)
        bd.dqkv, bd.dx_prev, bd.dprev_gamma, bd.scratch = P(o["dqkv"]), P(o["da"]), P(o["dlnp"]), P(scr_a)
        scr_t = torch.empty(lib.b4r_encoder_layer_bwd_scratch_floats(N), dtype=torch.float32, device=DEV)
        if layer_calls:
            _lib.check(lib.b4r_encoder_layer_fwd(C.byref(ad), C.byref(fd), stream()), "b4r_encoder_layer_fwd")
            _lib.check(lib.b4r_encoder_layer_bwd(C.byref(fd), C.byref(bd), P(o["dWo"]), P(o["dbo"]), P(o["dWqkv"]), P(o["dbqkv"]),
                                                 P(scr_t), stream()), "b4r_encoder_layer_bwd")
        else:
            _lib.check(lib.b4r_attn_block_fwd(C.byref(ad), stream()))
            _lib.check(lib.b4r_ffn_block_fwd(C.byref(fd), stream()))
            _lib.check(lib.b4r_ffn_block_bwd(C.byref(fd), stream()))
            _lib.check(lib.b4r_attn_block_bwd(C.byref(bd), stream()))
        torch.cuda.synchronize()
        return o

    whole, halves = run(True), run(False)
    # x1 not stored at all: the feed-forward kernels form it from z1 and the statistics (the attention epilogue's formula)
    lean = run(True, store_x1=False)
    for k in shapes:
        if k != "x1":
            assert T.maxdiff(lean[k], whole[k].cpu()) < 2e-6 * max(1.0, float(whole[k].abs().max())), k
    assert bool(torch.isnan(lean["x1"]).all())
    for k in shapes:
        if k not in ("dWo", "dbo", "dWqkv", "dbqkv"):
            assert torch.equal(whole[k], halves[k]), k
        assert bool(torch.isfinite(whole[k]).all()), k
    dz1 = whole["dz1"].cpu().double()
    if rate > 0:
        dz1 = dz1 * orc.dropout_keep_mask((N, 64), rate, seed, step, 2).double() / (1 - rate)
    ctx, x, dqkv = whole["ctx"].cpu().double(), t["x"].double(), whole["dqkv"].cpu().double()
    for got, want, what in ((whole["dWo"], ctx.T @ dz1, "dWo"), (whole["dbo"], dz1.sum(0), "dbo"), (whole["dWqkv"], x.T @ dqkv, "dWqkv"),
                            (whole["dbqkv"], dqkv.sum(0), "dbqkv")):
        assert T.maxdiff(got, want) < 3e-5 * max(1.0, float(want.abs().max())), what


@pytest.mark.parametrize("B,L,rate", [(4, 200, 0.2), (3, 37, 0.0)])
def test_attention_block_can_form_its_input_from_the_embedding_tables(B, L, rate):
    """First layer: b4r_attn_block_fwd with emb_ids runs the embedding stage (bert4rec_encoder.py:198-214: item row + position row ->
    LayerNorm -> dropout) itself.  Against b4r_embed_ln_fwd followed by the plain block: same dropout decisions, x / statistics and
    every block output to rounding (the two kernels sum the 64 columns of a row in different orders)."""
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    N, V, seed, step = B * L, 211, 19, 4
    t, mask = attn_inputs(B, L, seed=L + 3)
    g = {k: v.to(DEV) for k, v in t.items()}
    ids = torch.randint(-2, V + 2, (B, L), generator=torch.Generator().manual_seed(3)).to(DEV)   # a few out-of-range ids: row 0
    table, pos = rnd(V, 64, seed=8).to(DEV), (0.3 * rnd(L, 64, seed=9)).to(DEV)
    g0, be0 = (1.0 + 0.2 * rnd(64, seed=10)).to(DEV), (0.1 * rnd(64, seed=11)).to(DEV)
    st = T.new_state(seed, step) if rate > 0 else None
    maskd = mask.to(DEV)

    def run(embed):
        o = {k: torch.full(s_, float("nan"), dtype=torch.float32, device=DEV) for k, s_ in
             dict(x0=(N, 64), m0=(N,), r0=(N,), ctx=(N, 64), lse=(B, 2, L), z1=(N, 64), x1=(N, 64)).items()}
        bits = torch.zeros(lib.b4r_attn_keep_words(B, L, 2), dtype=torch.int32, device=DEV)
        d = _lib.AttnBlockDesc()
        d.B, d.L, d.H, d.heads, d.input_mask = B, L, 64, 2, P(maskd)
        d.Wqkv, d.bqkv, d.Wo, d.bo = P(g["Wqkv"]), P(g["bqkv"]), P(g["Wo"]), P(g["bo"])
        d.ln_gamma, d.ln_beta, d.ln_eps = P(g["g1"]), P(g["be1"]), 1e-12
        d.rng, d.probs_stream, d.probs_rate, d.out_stream, d.out_rate = P(st), 1, rate, 2, rate
        d.ctx, d.lse, d.keep_bits, d.z1, d.x1 = P(o["ctx"]), P(o["lse"]), P(bits), P(o["z1"]), P(o["x1"])
        if embed:
            d.emb_ids, d.emb_table, d.emb_pos, d.emb_gamma, d.emb_beta = P(ids), P(table), P(pos), P(g0), P(be0)
            d.emb_vocab, d.emb_eps, d.emb_stream, d.emb_rate = V, 1e-12, 0, rate
            d.emb_x, d.emb_mean, d.emb_rstd = P(o["x0"]), P(o["m0"]), P(o["r0"])
        else:
            _lib.check(lib.b4r_embed_ln_fwd(P(ids), B, L, P(table), V, P(pos), P(g0), P(be0), 64, 1e-12, P(o["x0"]), P(o["m0"]), P(o["r0"]),
                                            P(st), rate, stream()), "b4r_embed_ln_fwd")
            d.x = P(o["x0"])
        _lib.check(lib.b4r_attn_block_fwd(C.byref(d), stream()), "b4r_attn_block_fwd")
        torch.cuda.synchronize()
        return o, bits

    (a, bits_a), (b, bits_b) = run(True), run(False)
    assert torch.equal(a["x0"] == 0, b["x0"] == 0) and torch.equal(bits_a, bits_b)      # the same dropout decisions
    for k, tol in (("x0", 2e-6), ("m0", 1e-6), ("ctx", 2e-5), ("z1", 2e-5), ("x1", 2e-5), ("lse", 2e-5)):
        assert T.maxdiff(a[k], b[k].cpu()) < tol * max(1.0, float(b[k].abs().max())), k
    assert T.maxdiff(a["r0"] / b["r0"], torch.ones(N)) < 1e-5


# ---------------------------------------------------------------------------------------------------------------------------
# feed-forward half at hidden sizes 128 / 256 (b4r_ffn_wide_*: the reference's *_128.json / *_256.json configurations)
# ---------------------------------------------------------------------------------------------------------------------------
def wide_ffn_inputs(N, H, I, seed):
    return dict(x1=rnd(N, H, seed=seed + 1), W1=rnd(H, I, seed=seed + 4, scale=0.1), b1=0.1 * rnd(I, seed=seed + 5),
                W2=rnd(I, H, seed=seed + 6, scale=0.08), b2=0.1 * rnd(H, seed=seed + 7), g2=1.0 + 0.2 * rnd(H, seed=seed + 8),
                be2=0.1 * rnd(H, seed=seed + 9), dx2=rnd(N, H, seed=seed + 10))


def wide_ffn_reference(t, N, H, rate, seed, step, site, eps=1e-12):
    """fp64: fpre = x1 W1 + b1; f = gelu(fpre); x2 = LN2(x1 + drop(f W2 + b2)); loss = sum(x2 * dx2); autograd gradients."""
    d = {k: v.double().requires_grad_(k != "dx2") for k, v in t.items()}
    fpre = d["x1"] @ d["W1"] + d["b1"]
    fpre.retain_grad()
    f = gelu64(fpre)
    y = f @ d["W2"] + d["b2"]
    if rate > 0:
        y = y * orc.dropout_keep_mask((N, H), rate, seed, step, site).double() / (1.0 - rate)
    z2 = d["x1"] + y
    z2.retain_grad()
    x2, mean2, rstd2 = ln64(z2, d["g2"], d["be2"], eps)
    (x2 * d["dx2"]).sum().backward()
    return dict(fpre=fpre.detach(), f=f.detach(), z2=z2.detach(), x2=x2.detach(), mean2=mean2.detach(), rstd2=rstd2.detach(),
                dz2=z2.grad, df=fpre.grad, dx1=d["x1"].grad)


@pytest.mark.parametrize("H,I,N,rate", [(128, 512, 256, 0.0), (128, 512, 1000, 0.2), (128, 64, 31, 0.5), (256, 1024, 128, 0.0),
                                        (256, 1024, 777, 0.2), (256, 96, 200, 0.1), (128, 512, 51200 // 8 + 7, 0.2)])
def test_wide_ffn_block_matches_fp64_autograd(H, I, N, rate):
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_ffn_wide_supported(H, I) == 1
    seed, step, site = 977, 5, 9
    t = wide_ffn_inputs(N, H, I, seed=N + H)
    ref = wide_ffn_reference(t, N, H, rate, seed, step, site)
    g = {k: v.to(DEV) for k, v in t.items()}
    st = T.new_state(seed, step) if rate > 0 else None
    nan = float("nan")
    out = {k: torch.full(s, nan, dtype=torch.float32, device=DEV) for k, s in
           dict(z2=(N, H), x2=(N, H), mean2=(N,), rstd2=(N,), f=(N, I), fpre=(N, I), df=(N, I), dx1=(N, H)).items()}
    scratch = torch.empty(lib.b4r_ffn_wide_scratch_floats(H, I), dtype=torch.float32, device=DEV)
    d = _lib.FfnDesc()
    d.N, d.H, d.I = N, H, I
    d.x1, d.W1, d.b1, d.W2, d.b2 = P(g["x1"]), P(g["W1"]), P(g["b1"]), P(g["W2"]), P(g["b2"])
    d.ln_gamma, d.ln_beta, d.ln_eps = P(g["g2"]), P(g["be2"]), 1e-12
    d.rng, d.drop_stream, d.drop_rate = P(st), site, rate
    d.z2, d.x2, d.mean2, d.rstd2 = P(out["z2"]), P(out["x2"]), P(out["mean2"]), P(out["rstd2"])
    d.scratch = P(scratch)
    _lib.check(lib.b4r_ffn_wide_fwd(C.byref(d), P(out["f"]), P(out["fpre"]), stream()), "b4r_ffn_wide_fwd")
    torch.cuda.synchronize()
    for k in ("fpre", "f", "z2", "x2", "mean2"):   # 2e-5 of the largest entry (|z2| reaches 8 at inner size 1024), as the sums of the H = 64 test
        assert T.maxdiff(out[k], ref[k]) < max(1e-4, 2e-5 * float(ref[k].abs().max())), k
    assert T.maxdiff(out["rstd2"] / ref["rstd2"].float().to(DEV), torch.ones(N)) < 1e-4
    # the inference form (nothing of size [N, I] written) gives the same bits
    x2_keep = out["x2"].clone()
    out["x2"].fill_(nan)
    _lib.check(lib.b4r_ffn_wide_fwd(C.byref(d), None, None, stream()), "b4r_ffn_wide_fwd")
    torch.cuda.synchronize()
    assert torch.equal(out["x2"], x2_keep)

    dz2 = ref["dz2"].float().to(DEV)
    d.dz2 = P(dz2)
    for ready in (1, 0):
        out["df"].fill_(nan); out["dx1"].fill_(nan)
        _lib.check(lib.b4r_ffn_wide_bwd(C.byref(d), P(out["fpre"]), P(out["df"]), P(out["dx1"]), ready, stream()), "b4r_ffn_wide_bwd")
        torch.cuda.synchronize()
        scale = max(1.0, float(ref["dx1"].abs().max()))
        assert T.maxdiff(out["df"], ref["df"]) < 1e-4 * max(1.0, float(ref["df"].abs().max())), "df"
        assert T.maxdiff(out["dx1"], ref["dx1"]) < 1e-4 * scale, "dx1"


def test_wide_ffn_block_refuses_other_shapes():
    lib = _lib.load()
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
    assert lib.b4r_ffn_wide_supported(64, 256) == 0 and lib.b4r_ffn_wide_supported(128, 500) == 0
    d = _lib.FfnDesc()
    d.N, d.H, d.I = 16, 64, 256
    assert lib.b4r_ffn_wide_fwd(C.byref(d), None, None, stream()) == -2   # B4R_E_SHAPE
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_F32))
    assert lib.b4r_ffn_wide_supported(128, 512) == 0
    _lib.check(lib.b4r_set_gemm_mode(_lib.GEMM_BF16X3))
