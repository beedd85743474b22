"""Thin helpers for the GPU parity tests: call the C ABI (include/b4r.h) on torch device tensors."""
import ctypes as C

import torch

from bert4rec_amd import _lib
from bert4rec_amd._lib import GemmDesc, GemmTnDesc


def P(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def new_state(seed=1234, step=0, device="cuda"):
    st = torch.zeros(_lib.STATE_WORDS, dtype=torch.int32, device=device)
    st[_lib.ST_SEED] = seed
    st[_lib.ST_STEP_LO] = step
    st[_lib.ST_STEP:_lib.ST_STEP + 2].view(torch.int64)[0] = step
    return st


def state_floats(st):
    return st.cpu().view(torch.float32)


def gemm(A, B, M, N, K, b_is_nk=0, epi=_lib.EPI_NONE, bias=None, R=None, want_c2=False, qscale=1.0, qcols=0,
         rng=None, drop_stream=0, drop_rate=0.0, a_dropout=0, ldc=None, lda=None, ldb=None, c_pad_scratch=0):
    lib = _lib.load()
    ldc = ldc or N
    Cm = torch.full((M, ldc), float("nan"), dtype=torch.float32, device=A.device)
    C2 = torch.full((M, N), float("nan"), dtype=torch.float32, device=A.device) if want_c2 else None
    d = GemmDesc()
    d.A, d.lda = P(A), lda or A.stride(0)
    d.B, d.ldb = P(B), ldb or B.stride(0)
    d.C, d.ldc = P(Cm), ldc
    d.M, d.N, d.K = M, N, K
    d.b_is_nk, d.epilogue = b_is_nk, epi
    d.bias = P(bias)
    d.C2, d.ldc2 = P(C2), N
    d.R, d.ldr = P(R), (R.stride(0) if R is not None else 0)
    d.qscale, d.qcols = qscale, qcols
    d.rng, d.drop_stream, d.drop_rate, d.a_dropout = P(rng), drop_stream, drop_rate, a_dropout
    d.c_pad_scratch = c_pad_scratch
    _lib.check(lib.b4r_gemm_f32(C.byref(d), stream()), "b4r_gemm_f32")
    return Cm[:, :N], C2


def gemm_tn(A, B, R, Mo, No, want_colsum=False, want_colsum_a=False, rng=None, drop_stream=0, drop_rate=0.0, b_dropout=0):
    lib = _lib.load()
    out = torch.full((Mo, No), float("nan"), dtype=torch.float32, device=A.device)
    cs = torch.full((No,), float("nan"), dtype=torch.float32, device=A.device) if want_colsum else None
    csa = torch.full((Mo,), float("nan"), dtype=torch.float32, device=A.device) if want_colsum_a else None
    scratch = torch.empty(lib.b4r_gemm_tn_scratch_floats(R, Mo, No), dtype=torch.float32, device=A.device)
    d = GemmTnDesc()
    d.A, d.lda, d.B, d.ldb = P(A), A.stride(0), P(B), B.stride(0)
    d.out, d.ldo = P(out), No
    d.R, d.Mo, d.No = R, Mo, No
    d.colsum, d.colsum_a = P(cs), P(csa)
    d.rng, d.drop_stream, d.drop_rate, d.b_dropout, d.accumulate = P(rng), drop_stream, drop_rate, b_dropout, 0
    _lib.check(lib.b4r_gemm_tn_f32(C.byref(d), P(scratch), stream()), "b4r_gemm_tn_f32")
    return out, cs, csa


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())
