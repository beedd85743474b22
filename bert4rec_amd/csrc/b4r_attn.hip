// Keras MultiHeadAttention core (head_dim 32) on the exact-fp32 matrix cores, never materialising the [B,h,L,L] scores.
// These kernels serve the B4R_GEMM_F32 mode; the default mode runs the split-precision kernels of b4r_attn_rx.hip, which
// the C entry points at the bottom of this file dispatch to.
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
//
// LDS tiles are [rows][32 floats] with an XOR swizzle of the column (tile_idx) that makes BOTH operand access patterns
// conflict-free for ds_read_b32: (1) 16 rows x 2 adjacent columns (A operand of q.k^T-like products) and (2) 16 adjacent
// columns x rows 4g+s (A operand of the products that consume an accumulator tile).  A workgroup is 8 waves = 128 queries
// (or keys): K and V (52 KB at L = 200) are staged once per 128 rows and two workgroups fit a CU (16 waves).
#include "b4r_common.h"

namespace {

constexpr int WAVES = 8;         // waves per workgroup
constexpr int ROWS_WG = 16 * WAVES;

struct AttnP {
  const float* qkv; const int64_t* mask; const float* ctx; const float* lse_in; const float* dctx;
  float* ctx_out; float* lse_out; float* dqkv;
  int B, L, heads, H, Lp;
  int reg_rows;  // rows per LDS region of the dQ kernel: max(Lp, 128)
  float qscale;
  DropArgs drop;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// swizzled position of (row, col) in a [rows][32] tile; bit 0 of the column is untouched (8-byte pairs stay adjacent)
__device__ __forceinline__ int tile_idx(int row, int col) {
  return row * 32 + (col ^ ((2 * (row & 15)) ^ (16 * ((row >> 2) & 1))));
}

// rows [0,nrows) of a [*,32] head slice -> swizzled LDS tile; rows beyond `valid` are zero
__device__ __forceinline__ void load_head_rows(float* dst, const float* src, int64_t row0, int ld, int nrows, int valid) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  for (int f = threadIdx.x; f < nrows * 8; f += 64 * WAVES) {
    const int r = f >> 3, c = (f & 7) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < valid) v = *reinterpret_cast<const f32x4*>(src + (row0 + r) * ld + c);
    *reinterpret_cast<f32x2*>(dst + tile_idx(r, c)) = (f32x2){v[0], v[1]};
    *reinterpret_cast<f32x2*>(dst + tile_idx(r, c + 2)) = (f32x2){v[2], v[3]};
  }
}

// D[r] = sum_c dO[r][c] * O[r][c] over the 32 columns of this head, for rows [0,nrows); rows beyond valid -> 0
__device__ __forceinline__ void rowdot_head(float* sD, const float* dO, const float* O, int64_t row0, int ld, int nrows, int valid) {
  for (int base = 0; base < nrows; base += 16 * WAVES) {
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
// forward: workgroup = 128 queries of one (batch, head); wave = 16 queries x all keys
// LDS: [K tile | V tile | sAdd]; the query tile borrows the K region first (its fragments live in registers afterwards)
// -----------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(64 * WAVES) void attn_fwd_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = KT * 16;
  float* sK = smem;
  float* sV = sK + Lp * 32;
  float* sAdd = sV + Lp * 32;

  const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * ROWS_WG;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const float amax = b4r_seq_amax(p.mask + row0, L);   // all threads, before any early exit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;

  load_head_rows(sK, p.qkv + hd * 32, row0 + q0, ld3, ROWS_WG, L - q0);   // Q tile in the K region
  __syncthreads();
  float qf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) qf[s] = sK[tile_idx(16 * wave + i, 4 * s + g)];
  __syncthreads();
  load_head_rows(sK, p.qkv + H + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sV, p.qkv + 2 * H + hd * 32, row0, ld3, Lp, L);
  for (int k = threadIdx.x; k < Lp; k += 64 * WAVES)
    sAdd[k] = (k < L) ? (1.0f - (float)p.mask[row0 + k]) * -1e9f : -INFINITY;
  __syncthreads();
  if (q0 + 16 * wave >= L) return;  // wave-uniform; no barrier below

  const int q = q0 + 16 * wave + i;
  f32x4 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) c = mfma16(sK[tile_idx(16 * t + i, 4 * s + g)], qf[s], c);
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
  if (g == 0 && q < L && p.lse_out) p.lse_out[((int64_t)b * p.heads + hd) * L + q] = (m - amax) + __logf(sum);

  DropCtx dctx = b4r_drop_ctx(p.drop);
  const uint64_t dbase = (((uint64_t)b * p.heads + hd) * L + (uint64_t)(q < L ? q : 0)) * (uint64_t)B4R_ATTN_PITCH;
  f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const uint32_t k4 = dctx.on ? b4r_keep4(dctx, dbase + (uint64_t)(16 * t + 4 * g)) : 15u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int key = 16 * t + 4 * g + s;
      float pv = acc[t][s] * inv;
      if (dctx.on) pv = ((k4 >> s) & 1u) ? pv * dctx.scale : 0.f;
      o0 = mfma16(sV[tile_idx(key, i)], pv, o0);
      o1 = mfma16(sV[tile_idx(key, 16 + i)], pv, o1);
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
// LDS: [K tile | V tile | sAdd | sD]; Q and dO tiles borrow the K / V regions first
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_dq_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = p.Lp, KT = Lp / 16;
  float* sK = smem;
  float* sV = sK + p.reg_rows * 32;
  float* sAdd = sV + p.reg_rows * 32;
  float* sD = sAdd + p.reg_rows;

  const int b = blockIdx.z, hd = blockIdx.y, q0 = blockIdx.x * ROWS_WG;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const float amax = b4r_seq_amax(p.mask + row0, L);   // all threads, before any early exit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;

  load_head_rows(sK, p.qkv + hd * 32, row0 + q0, ld3, ROWS_WG, L - q0);   // Q tile
  load_head_rows(sV, p.dctx + hd * 32, row0 + q0, H, ROWS_WG, L - q0);    // dO tile
  rowdot_head(sD, p.dctx + hd * 32, p.ctx + hd * 32, row0 + q0, H, ROWS_WG, L - q0);
  __syncthreads();
  float qf[8], dof[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    qf[s] = sK[tile_idx(16 * wave + i, 4 * s + g)];
    dof[s] = sV[tile_idx(16 * wave + i, 4 * s + g)];
  }
  const float Dq = sD[16 * wave + i];
  __syncthreads();
  load_head_rows(sK, p.qkv + H + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sV, p.qkv + 2 * H + hd * 32, row0, ld3, Lp, L);
  for (int k = threadIdx.x; k < Lp; k += 64 * WAVES)
    sAdd[k] = (k < L) ? (1.0f - (float)p.mask[row0 + k]) * -1e9f : -INFINITY;
  __syncthreads();
  if (q0 + 16 * wave >= L) return;

  const int q = q0 + 16 * wave + i;
  const bool qlive = q < L;
  const float lse = qlive ? p.lse_in[((int64_t)b * p.heads + hd) * L + q] : 0.f;
  DropCtx dctx = b4r_drop_ctx(p.drop);
  const uint64_t dbase = (((uint64_t)b * p.heads + hd) * L + (uint64_t)(qlive ? q : 0)) * (uint64_t)B4R_ATTN_PITCH;

  f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < KT; ++t) {
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      sc = mfma16(sK[tile_idx(16 * t + i, 4 * s + g)], qf[s], sc);
      da = mfma16(sV[tile_idx(16 * t + i, 4 * s + g)], dof[s], da);
    }
    const f32x4 ad = *reinterpret_cast<const f32x4*>(&sAdd[16 * t + 4 * g]);
    const uint32_t k4 = dctx.on ? b4r_keep4(dctx, dbase + (uint64_t)(16 * t + 4 * g)) : 15u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int key = 16 * t + 4 * g + s;
      const float pr = __expf(((sc[s] + ad[s]) - amax) - lse);
      float dA = da[s];
      if (dctx.on) dA = ((k4 >> s) & 1u) ? dA * dctx.scale : 0.f;
      const float ds = pr * (dA - Dq);
      dq0 = mfma16(sK[tile_idx(key, i)], ds, dq0);
      dq1 = mfma16(sK[tile_idx(key, 16 + i)], ds, dq1);
    }
  }
  if (qlive) {
    float* o = p.dqkv + (row0 + q) * ld3 + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(o) = dq0 * p.qscale;
    *reinterpret_cast<f32x4*>(o + 16) = dq1 * p.qscale;
  }
}

// -----------------------------------------------------------------------------------------------------------
// backward, dK / dV: workgroup = 128 keys of one (batch, head); wave = 16 keys x all queries
// LDS: [Q tile | dO tile | sLse | sD]
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_dkv_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = p.Lp, KT = Lp / 16;
  float* sQ = smem;
  float* sdO = sQ + Lp * 32;
  float* sLse = sdO + Lp * 32;
  float* sD = sLse + Lp;

  const int b = blockIdx.z, hd = blockIdx.y;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const float amax = b4r_seq_amax(p.mask + row0, L);   // all threads, before any early exit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;

  load_head_rows(sQ, p.qkv + hd * 32, row0, ld3, Lp, L);
  load_head_rows(sdO, p.dctx + hd * 32, row0, H, Lp, L);
  rowdot_head(sD, p.dctx + hd * 32, p.ctx + hd * 32, row0, H, Lp, L);
  for (int k = threadIdx.x; k < Lp; k += 64 * WAVES)
    sLse[k] = (k < L) ? p.lse_in[((int64_t)b * p.heads + hd) * L + k] : INFINITY;  // +inf => probability 0 for pad queries
  __syncthreads();

  const int k0 = (blockIdx.x * WAVES + wave) * 16;
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
      sc = mfma16(sQ[tile_idx(16 * t + i, 4 * s + g)], kf[s], sc);
      da = mfma16(sdO[tile_idx(16 * t + i, 4 * s + g)], vf[s], da);
    }
    const f32x4 ls = *reinterpret_cast<const f32x4*>(&sLse[16 * t + 4 * g]);
    const f32x4 dd = *reinterpret_cast<const f32x4*>(&sD[16 * t + 4 * g]);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int qq = 16 * t + 4 * g + s;
      const float pr = __expf(((sc[s] + add) - amax) - ls[s]);
      float ad = pr, dA = da[s];
      if (dctx.on) {
        const bool keep = b4r_keep(dctx, (hbase + (uint64_t)(qq < L ? qq : 0)) * (uint64_t)B4R_ATTN_PITCH + (uint64_t)(klive ? key : 0));
        ad = keep ? pr * dctx.scale : 0.f;
        dA = keep ? dA * dctx.scale : 0.f;
      }
      const float ds = pr * (dA - dd[s]);
      dv0 = mfma16(sdO[tile_idx(qq, i)], ad, dv0);
      dv1 = mfma16(sdO[tile_idx(qq, 16 + i)], ad, dv1);
      dk0 = mfma16(sQ[tile_idx(qq, i)], ds, dk0);
      dk1 = mfma16(sQ[tile_idx(qq, 16 + i)], ds, dk1);
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
int set_lds(K kernel, size_t bytes) { return b4r_raise_lds((const void*)kernel, bytes, "attention"); }

int check_common(const char* who, const float* qkv, const int64_t* mask, int B, int L, int heads) {
  B4R_CHECK_ARG(qkv && mask, B4R_E_BADARG, "%s: null argument", who);
  B4R_CHECK_ARG(B > 0 && L > 0 && heads > 0, B4R_E_SHAPE, "%s: bad shape", who);
  B4R_CHECK_ARG(key_tiles(L) != 0, B4R_E_SHAPE, "%s: sequence length %d > 256 is not supported", who, L);
  B4R_CHECK_ARG(b4r_aligned16(qkv), B4R_E_ALIGN, "%s: qkv must be 16-byte aligned", who);
  return B4R_OK;
}

inline int tile_rows(int Lp) { return Lp > ROWS_WG ? Lp : ROWS_WG; }  // a region must also hold a 128-row Q / dO tile

// the dropout decisions are only stored / consumed by the split-precision kernels; the fp32 kernels hash again
int check_bits(const char* who, const DropArgs& drop, const uint32_t* keep_bits) {
  if (drop.rng == nullptr) return B4R_OK;
  B4R_CHECK_ARG(keep_bits != nullptr, B4R_E_BADARG, "%s: dropout in the bf16x3 mode needs the keep_bits buffer (b4r_attn_keep_words)", who);
  B4R_CHECK_ARG(b4r_aligned16(keep_bits), B4R_E_ALIGN, "%s: keep_bits must be 16-byte aligned", who);
  return B4R_OK;
}

}  // namespace

int64_t b4r_attn_rx_keep_words(int B, int L, int heads);
int b4r_attn_rx_fwd_launch(const float* qkv, const int64_t* mask, int B, int L, int heads, float* ctx, float* lse,
                           const DropArgs& drop, uint32_t* keep_bits, hipStream_t stream);
int b4r_attn_rx_bwd_launch(const float* qkv, const int64_t* mask, const float* ctx, const float* lse, const float* dctx,
                           int B, int L, int heads, float qscale, float* dqkv, const DropArgs& drop,
                           const uint32_t* keep_bits, hipStream_t stream, hipStream_t stream_dkv);

int64_t b4r_attn32_keep_words(int32_t B, int32_t L, int32_t heads);
extern "C" int64_t b4r_attn_keep_words(int32_t B, int32_t L, int32_t heads) {
  if (B <= 0 || L <= 0 || heads <= 0) return 0;
  // round 1's layout (b4r_attn_rx.hip), then one word per (query, 32-key tile) for b4r_attn32.hip's backward
  return b4r_attn_rx_keep_words(B, L, heads) + b4r_attn32_keep_words(B, L, heads);
}

extern "C" int b4r_attn_fwd(const float* qkv, const int64_t* input_mask, int32_t B, int32_t L, int32_t heads, float* ctx,
                            float* lse, const uint32_t* rng, uint32_t drop_stream, float drop_rate, uint32_t* keep_bits,
                            b4r_stream_t stream) {
  int rc = check_common("b4r_attn_fwd", qkv, input_mask, B, L, heads);
  if (rc) return rc;
  B4R_CHECK_ARG(ctx != nullptr, B4R_E_BADARG, "b4r_attn_fwd: null ctx");
  if (b4r_get_gemm_mode() == B4R_GEMM_BF16X3) {
    const DropArgs drop = b4r_make_drop(rng, drop_stream, drop_rate, 1);
    rc = check_bits("b4r_attn_fwd", drop, keep_bits);
    if (rc) return rc;
    return b4r_attn_rx_fwd_launch(qkv, input_mask, B, L, heads, ctx, lse, drop, keep_bits, (hipStream_t)stream);
  }
  AttnP p{};
  p.qkv = qkv; p.mask = input_mask; p.ctx_out = ctx; p.lse_out = lse;
  p.B = B; p.L = L; p.heads = heads; p.H = heads * 32;
  const int KT = key_tiles(L);
  p.Lp = KT * 16;
  p.drop = b4r_make_drop(rng, drop_stream, drop_rate, 1);
  // the K region must hold the 128-row query tile too; regions are laid out with stride Lp*32, so for short sequences the
  // query tile spills into the V region, which is only filled after the query fragments have been read
  const size_t sh = ((size_t)2 * p.Lp * 32 + p.Lp + (p.Lp < ROWS_WG ? (size_t)(ROWS_WG - p.Lp) * 32 : 0)) * sizeof(float);
  dim3 grid(b4r_cdiv(L, ROWS_WG), heads, B);
#define FWD_CASE(KT_)                                                                                   \
  case KT_:                                                                                             \
    rc = set_lds(attn_fwd_kernel<KT_>, sh);                                                             \
    if (rc) return rc;                                                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<KT_>), grid, dim3(64 * WAVES), sh, (hipStream_t)stream, p);     \
    break;
  switch (KT) {
    FWD_CASE(4) FWD_CASE(8) FWD_CASE(13) FWD_CASE(16)
    default: b4r_set_error("b4r_attn_fwd: internal"); return B4R_E_SHAPE;
  }
#undef FWD_CASE
  B4R_CHECK_LAUNCH("b4r_attn_fwd");
  return B4R_OK;
}

int b4r_attn_bwd_streams(const float* qkv, const int64_t* input_mask, const float* ctx, const float* lse, const float* dctx,
                         int32_t B, int32_t L, int32_t heads, float qscale, float* dqkv, const uint32_t* rng,
                         uint32_t drop_stream, float drop_rate, const uint32_t* keep_bits, hipStream_t stream,
                         hipStream_t stream_dkv);

extern "C" int b4r_attn_bwd(const float* qkv, const int64_t* input_mask, const float* ctx, const float* lse,
                            const float* dctx, int32_t B, int32_t L, int32_t heads, float qscale, float* dqkv,
                            const uint32_t* rng, uint32_t drop_stream, float drop_rate, const uint32_t* keep_bits,
                            b4r_stream_t stream) {
  return b4r_attn_bwd_streams(qkv, input_mask, ctx, lse, dctx, B, L, heads, qscale, dqkv, rng, drop_stream, drop_rate,
                              keep_bits, (hipStream_t)stream, (hipStream_t)stream);
}

// stream_dkv: where the dK/dV kernel goes (the backward pass overlaps it with dQ); the caller orders the two streams
int b4r_attn_bwd_streams(const float* qkv, const int64_t* input_mask, const float* ctx, const float* lse, const float* dctx,
                         int32_t B, int32_t L, int32_t heads, float qscale, float* dqkv, const uint32_t* rng,
                         uint32_t drop_stream, float drop_rate, const uint32_t* keep_bits, hipStream_t stream,
                         hipStream_t stream_dkv) {
  int rc = check_common("b4r_attn_bwd", qkv, input_mask, B, L, heads);
  if (rc) return rc;
  B4R_CHECK_ARG(ctx && lse && dctx && dqkv, B4R_E_BADARG, "b4r_attn_bwd: null argument");
  B4R_CHECK_ARG(b4r_aligned16(ctx) && b4r_aligned16(dctx) && b4r_aligned16(dqkv), B4R_E_ALIGN, "b4r_attn_bwd: operands must be 16-byte aligned");
  if (b4r_get_gemm_mode() == B4R_GEMM_BF16X3) {
    const DropArgs drop = b4r_make_drop(rng, drop_stream, drop_rate, 1);
    rc = check_bits("b4r_attn_bwd", drop, keep_bits);
    if (rc) return rc;
    return b4r_attn_rx_bwd_launch(qkv, input_mask, ctx, lse, dctx, B, L, heads, qscale, dqkv, drop, keep_bits, stream, stream_dkv);
  }
  AttnP p{};
  p.qkv = qkv; p.mask = input_mask; p.ctx = ctx; p.lse_in = lse; p.dctx = dctx; p.dqkv = dqkv;
  p.B = B; p.L = L; p.heads = heads; p.H = heads * 32; p.qscale = qscale;
  p.Lp = key_tiles(L) * 16;
  p.drop = b4r_make_drop(rng, drop_stream, drop_rate, 1);
  dim3 grid(b4r_cdiv(L, ROWS_WG), heads, B);
  // dQ: the K and V regions must each also hold a 128-row Q / dO tile
  p.reg_rows = tile_rows(p.Lp);
  const size_t sh_dq = ((size_t)2 * p.reg_rows * 32 + p.reg_rows + ROWS_WG) * sizeof(float);
  rc = set_lds(attn_bwd_dq_kernel, sh_dq);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(64 * WAVES), sh_dq, (hipStream_t)stream, p);
  B4R_CHECK_LAUNCH("b4r_attn_bwd dq");
  const size_t sh_kv = ((size_t)2 * p.Lp * 32 + 2 * p.Lp) * sizeof(float);
  rc = set_lds(attn_bwd_dkv_kernel, sh_kv);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(64 * WAVES), sh_kv, stream_dkv, p);
  B4R_CHECK_LAUNCH("b4r_attn_bwd dkv");
  return B4R_OK;
}
