// Keras MultiHeadAttention core (head_dim 32) on the exact-fp32 matrix cores, never materialising the [B,h,L,L] scores.
//
//   scores = q k^T + (1 - input_mask[b,key]) * -1e9 ; A = softmax(scores) ; A = dropout(A) ; ctx = A v
//   (tfm SelfAttentionMask is key-padding only: bert4rec_encoder.py:134-135,216; q arrives pre-scaled by 1/sqrt(d))
//
// v_mfma_f32_16x16x4_f32 maps (guide §3): lane l holds A[row=l&15][k=l>>4], B[k=l>>4][col=l&15];
// D: col = l&15, row = 4*(l>>4) + reg.
//
// Orientation trick: the forward and the dQ kernel compute the TRANSPOSED score tile S^T = K.Q^T, so a lane's 4 accumulator
// registers are 4 consecutive KEYS of one query.  Row (per-query) softmax statistics are then register reductions plus two
// xor-shuffles, and the probability tile is already laid out as the B operand of the next product (O^T = V^T.P^T sums over
// the accumulator's ROW index), so nothing is transposed through LDS.  The dK/dV kernel uses the other orientation
// (S = Q.K^T, a wave owns 16 keys and sweeps the queries) for the same reason: dV^T = dO^T.A and dK^T = Q^T.dS sum over
// queries = its accumulator rows.  No cross-workgroup sums, so the backward is bitwise reproducible.
#include "b4r_common.h"

namespace {

constexpr int LDH = 36;  // LDS row stride (floats) of the [rows][32] operand tiles: 16-B aligned rows, <=2-way conflicts

struct AttnP {
  const float* qkv; const int64_t* mask; const float* ctx; const float* lse_in; const float* dctx;
  float* ctx_out; float* lse_out; float* dqkv;
  int B, L, heads, H, Lp;
  float qscale;
  DropArgs drop;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// rows [0,nrows) of a [*,32] head slice -> LDS tile with stride LDH; rows beyond `valid` are zero
__device__ __forceinline__ void load_head_rows(float* dst, const float* src, int64_t row0, int ld, int nrows, int valid) {
  for (int f = threadIdx.x; f < nrows * 8; f += 256) {
    const int r = f >> 3, c = (f & 7) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < valid) v = *reinterpret_cast<const f32x4*>(src + (row0 + r) * ld + c);
    *reinterpret_cast<f32x4*>(dst + r * LDH + c) = v;
  }
}

// D[r] = sum_c dO[r][c] * O[r][c] over the 32 columns of this head, for rows [0,nrows); rows beyond valid -> 0
__device__ __forceinline__ void rowdot_head(float* sD, const float* dO, const float* O, int64_t row0, int ld, int nrows, int valid) {
  for (int base = 0; base < nrows; base += 64) {
    const int r = base + (threadIdx.x >> 2), part = threadIdx.x & 3;
    float s = 0.f;
    if (r < valid) {
      const float* a = dO + (row0 + r) * ld + part * 8;
      const float* b = O + (row0 + r) * ld + part * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(a), a1 = *reinterpret_cast<const f32x4*>(a + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b), b1 = *reinterpret_cast<const f32x4*>(b + 4);
      s = (a0[0] * b0[0] + a0[1] * b0[1]) + (a0[2] * b0[2] + a0[3] * b0[3]) +
          (a1[0] * b1[0] + a1[1] * b1[1]) + (a1[2] * b1[2] + a1[3] * b1[3]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0 && r < nrows) sD[r] = s;
  }
}

// -----------------------------------------------------------------------------------------------------------
// forward: workgroup = 64 queries of one (batch, head); wave = 16 queries x all keys
// -----------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = KT * 16;
  float* sK = smem;
  float* sV = sK + Lp * LDH;
  float* sQ = sV + Lp * LDH;
  float* sAdd = sQ + 64 * LDH;

  const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 64;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;

  load_head_rows(sK, p.qkv + H + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sV, p.qkv + 2 * H + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sQ, p.qkv + hd * 32, row0 + q0, ld3, 64, L - q0);
  for (int k = threadIdx.x; k < Lp; k += 256)
    sAdd[k] = (k < L) ? (1.0f - (float)p.mask[row0 + k]) * -1e9f : -INFINITY;
  __syncthreads();

  const int q = q0 + 16 * wave + i;
  float qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s] = sQ[(16 * wave + i) * LDH + 4 * s + g];

  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) c = mfma16(sK[(16 * t + i) * LDH + 4 * s + g], qf[s], c);
    acc[t] = c;
  }
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const f32x4 ad = *reinterpret_cast<const f32x4*>(&sAdd[16 * t + 4 * g]);
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[t][r] += ad[r]; m = fmaxf(m, acc[t][r]); }
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const float e = __expf(acc[t][r] - m); acc[t][r] = e; sum += e; }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (g == 0 && q < L && p.lse_out) p.lse_out[((int64_t)b * p.heads + hd) * L + q] = m + __logf(sum);

  DropCtx dctx = b4r_drop_ctx(p.drop);
  const uint64_t dbase = (((uint64_t)b * p.heads + hd) * L + (uint64_t)(q < L ? q : 0)) * (uint64_t)L;
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KT; ++t) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int key = 16 * t + 4 * g + s;
      float pv = acc[t][s] * inv;
      if (dctx.on) pv = b4r_keep(dctx, dbase + (uint64_t)key) ? pv * dctx.scale : 0.f;
      o0 = mfma16(sV[key * LDH + i], pv, o0);
      o1 = mfma16(sV[key * LDH + 16 + i], pv, o1);
    }
  }
  if (q < L) {
    float* o = p.ctx_out + (row0 + q) * H + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(o) = o0;
    *reinterpret_cast<f32x4*>(o + 16) = o1;
  }
}

// -----------------------------------------------------------------------------------------------------------
// backward, dQ: same decomposition as the forward; probabilities recomputed from the saved log-sum-exp
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = p.Lp, KT = Lp / 16;
  float* sK = smem;
  float* sV = sK + Lp * LDH;
  float* sQ = sV + Lp * LDH;
  float* sdO = sQ + 64 * LDH;
  float* sAdd = sdO + 64 * LDH;
  float* sD = sAdd + Lp;

  const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * 64;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;

  load_head_rows(sK, p.qkv + H + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sV, p.qkv + 2 * H + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sQ, p.qkv + hd * 32, row0 + q0, ld3, 64, L - q0);
  load_head_rows(sdO, p.dctx + hd * 32, row0 + q0, H, 64, L - q0);
  rowdot_head(sD, p.dctx + hd * 32, p.ctx + hd * 32, row0 + q0, H, 64, L - q0);
  for (int k = threadIdx.x; k < Lp; k += 256)
    sAdd[k] = (k < L) ? (1.0f - (float)p.mask[row0 + k]) * -1e9f : -INFINITY;
  __syncthreads();

  const int q = q0 + 16 * wave + i;
  const bool qlive = q < L;
  const float lse = qlive ? p.lse_in[((int64_t)b * p.heads + hd) * L + q] : 0.f;
  const float Dq = sD[16 * wave + i];
  float qf[8], dof[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    qf[s] = sQ[(16 * wave + i) * LDH + 4 * s + g];
    dof[s] = sdO[(16 * wave + i) * LDH + 4 * s + g];
  }
  DropCtx dctx = b4r_drop_ctx(p.drop);
  const uint64_t dbase = (((uint64_t)b * p.heads + hd) * L + (uint64_t)(qlive ? q : 0)) * (uint64_t)L;

  f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < KT; ++t) {
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      sc = mfma16(sK[(16 * t + i) * LDH + 4 * s + g], qf[s], sc);
      da = mfma16(sV[(16 * t + i) * LDH + 4 * s + g], dof[s], da);
    }
    const f32x4 ad = *reinterpret_cast<const f32x4*>(&sAdd[16 * t + 4 * g]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int key = 16 * t + 4 * g + s;
      const float pr = __expf(sc[s] + ad[s] - lse);
      float dA = da[s];
      if (dctx.on) dA = b4r_keep(dctx, dbase + (uint64_t)key) ? dA * dctx.scale : 0.f;
      const float ds = pr * (dA - Dq);
      dq0 = mfma16(sK[key * LDH + i], ds, dq0);
      dq1 = mfma16(sK[key * LDH + 16 + i], ds, dq1);
    }
  }
  if (qlive) {
    float* o = p.dqkv + (row0 + q) * ld3 + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(o) = dq0 * p.qscale;
    *reinterpret_cast<f32x4*>(o + 16) = dq1 * p.qscale;
  }
}

// -----------------------------------------------------------------------------------------------------------
// backward, dK / dV: workgroup = 64 keys of one (batch, head); wave = 16 keys x all queries
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = p.Lp, KT = Lp / 16;
  float* sQ = smem;
  float* sdO = sQ + Lp * LDH;
  float* sLse = sdO + Lp * LDH;
  float* sD = sLse + Lp;

  const int b = blockIdx.z, hd = blockIdx.y;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;

  load_head_rows(sQ, p.qkv + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sdO, p.dctx + hd * 32, row0, H, Lp, L);
  rowdot_head(sD, p.dctx + hd * 32, p.ctx + hd * 32, row0, H, Lp, L);
  for (int k = threadIdx.x; k < Lp; k += 256)
    sLse[k] = (k < L) ? p.lse_in[((int64_t)b * p.heads + hd) * L + k] : INFINITY;  // +inf => probability 0 for pad queries
  __syncthreads();

  const int k0 = (blockIdx.x * 4 + wave) * 16;
  if (k0 >= L) return;  // no barrier below
  const int key = k0 + i;
  const bool klive = key < L;
  float kf[8], vf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    kf[s] = klive ? p.qkv[(row0 + key) * ld3 + H + hd * 32 + 4 * s + g] : 0.f;
    vf[s] = klive ? p.qkv[(row0 + key) * ld3 + 2 * H + hd * 32 + 4 * s + g] : 0.f;
  }
  const float add = klive ? (1.0f - (float)p.mask[row0 + key]) * -1e9f : -INFINITY;
  DropCtx dctx = b4r_drop_ctx(p.drop);
  const uint64_t hbase = ((uint64_t)b * p.heads + hd) * (uint64_t)L;

  f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = {0.f, 0.f, 0.f, 0.f}, dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < KT; ++t) {
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      sc = mfma16(sQ[(16 * t + i) * LDH + 4 * s + g], kf[s], sc);
      da = mfma16(sdO[(16 * t + i) * LDH + 4 * s + g], vf[s], da);
    }
    const f32x4 ls = *reinterpret_cast<const f32x4*>(&sLse[16 * t + 4 * g]);
    const f32x4 dd = *reinterpret_cast<const f32x4*>(&sD[16 * t + 4 * g]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int qq = 16 * t + 4 * g + s;
      const float pr = __expf(sc[s] + add - ls[s]);
      float ad = pr, dA = da[s];
      if (dctx.on) {
        const bool keep = b4r_keep(dctx, (hbase + (uint64_t)(qq < L ? qq : 0)) * (uint64_t)L + (uint64_t)(klive ? key : 0));
        ad = keep ? pr * dctx.scale : 0.f;
        dA = keep ? dA * dctx.scale : 0.f;
      }
      const float ds = pr * (dA - dd[s]);
      dv0 = mfma16(sdO[qq * LDH + i], ad, dv0);
      dv1 = mfma16(sdO[qq * LDH + 16 + i], ad, dv1);
      dk0 = mfma16(sQ[qq * LDH + i], ds, dk0);
      dk1 = mfma16(sQ[qq * LDH + 16 + i], ds, dk1);
    }
  }
  if (klive) {
    float* ok = p.dqkv + (row0 + key) * ld3 + H + hd * 32 + 4 * g;
    float* ov = p.dqkv + (row0 + key) * ld3 + 2 * H + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(ok) = dk0;
    *reinterpret_cast<f32x4*>(ok + 16) = dk1;
    *reinterpret_cast<f32x4*>(ov) = dv0;
    *reinterpret_cast<f32x4*>(ov + 16) = dv1;
  }
}

int key_tiles(int L) {
  if (L <= 64) return 4;
  if (L <= 128) return 8;
  if (L <= 208) return 13;
  if (L <= 256) return 16;
  return 0;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { b4r_set_error("attention: cannot raise the LDS limit to %zu: %s", bytes, hipGetErrorString(e)); return B4R_E_HIP; }
  }
  return B4R_OK;
}

int check_common(const char* who, const float* qkv, const int64_t* mask, int B, int L, int heads) {
  B4R_CHECK_ARG(qkv && mask, B4R_E_BADARG, "%s: null argument", who);
  B4R_CHECK_ARG(B > 0 && L > 0 && heads > 0, B4R_E_SHAPE, "%s: bad shape", who);
  B4R_CHECK_ARG(key_tiles(L) != 0, B4R_E_SHAPE, "%s: sequence length %d > 256 is not supported", who, L);
  B4R_CHECK_ARG(b4r_aligned16(qkv), B4R_E_ALIGN, "%s: qkv must be 16-byte aligned", who);
  return B4R_OK;
}

}  // namespace

extern "C" int b4r_attn_fwd(const float* qkv, const int64_t* input_mask, int32_t B, int32_t L, int32_t heads, float* ctx,
                            float* lse, const uint32_t* rng, uint32_t drop_stream, float drop_rate, b4r_stream_t stream) {
  int rc = check_common("b4r_attn_fwd", qkv, input_mask, B, L, heads);
  if (rc) return rc;
  B4R_CHECK_ARG(ctx != nullptr, B4R_E_BADARG, "b4r_attn_fwd: null ctx");
  AttnP p{};
  p.qkv = qkv; p.mask = input_mask; p.ctx_out = ctx; p.lse_out = lse;
  p.B = B; p.L = L; p.heads = heads; p.H = heads * 32;
  const int KT = key_tiles(L);
  p.Lp = KT * 16;
  p.drop = b4r_make_drop(rng, drop_stream, drop_rate, 1);
  const size_t sh = ((size_t)2 * p.Lp * LDH + 64 * LDH + p.Lp) * sizeof(float);
  dim3 grid(b4r_cdiv(L, 64), heads, B);
#define FWD_CASE(KT_)                                                                                   \
  case KT_:                                                                                             \
    rc = set_lds(attn_fwd_kernel<KT_>, sh);                                                             \
    if (rc) return rc;                                                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<KT_>), grid, dim3(256), sh, (hipStream_t)stream, p);            \
    break;
  switch (KT) {
    FWD_CASE(4) FWD_CASE(8) FWD_CASE(13) FWD_CASE(16)
    default: b4r_set_error("b4r_attn_fwd: internal"); return B4R_E_SHAPE;
  }
#undef FWD_CASE
  B4R_CHECK_LAUNCH("b4r_attn_fwd");
  return B4R_OK;
}

extern "C" int b4r_attn_bwd(const float* qkv, const int64_t* input_mask, const float* ctx, const float* lse,
                            const float* dctx, int32_t B, int32_t L, int32_t heads, float qscale, float* dqkv,
                            const uint32_t* rng, uint32_t drop_stream, float drop_rate, b4r_stream_t stream) {
  int rc = check_common("b4r_attn_bwd", qkv, input_mask, B, L, heads);
  if (rc) return rc;
  B4R_CHECK_ARG(ctx && lse && dctx && dqkv, B4R_E_BADARG, "b4r_attn_bwd: null argument");
  AttnP p{};
  p.qkv = qkv; p.mask = input_mask; p.ctx = ctx; p.lse_in = lse; p.dctx = dctx; p.dqkv = dqkv;
  p.B = B; p.L = L; p.heads = heads; p.H = heads * 32; p.qscale = qscale;
  p.Lp = key_tiles(L) * 16;
  p.drop = b4r_make_drop(rng, drop_stream, drop_rate, 1);
  dim3 grid(b4r_cdiv(L, 64), heads, B);
  const size_t sh_dq = ((size_t)2 * p.Lp * LDH + 2 * 64 * LDH + p.Lp + 64) * sizeof(float);
  rc = set_lds(attn_bwd_dq_kernel, sh_dq);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), sh_dq, (hipStream_t)stream, p);
  B4R_CHECK_LAUNCH("b4r_attn_bwd dq");
  const size_t sh_kv = ((size_t)2 * p.Lp * LDH + 2 * p.Lp) * sizeof(float);
  rc = set_lds(attn_bwd_dkv_kernel, sh_kv);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), sh_kv, (hipStream_t)stream, p);
  B4R_CHECK_LAUNCH("b4r_attn_bwd dkv");
  return B4R_OK;
}
