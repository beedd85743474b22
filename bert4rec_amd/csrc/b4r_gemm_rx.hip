// Dense layers on the bf16 matrix cores with a 3-term split ("bf16x3"): rx_gemm_kn_kernel / rx_gemm_nk_kernel for K = 32
// or 64 (every projection that reduces over the hidden size of the 64-wide configurations: QKV, attention output, FFN-in,
// the MLM transform, the materialising vocabulary projection, and the input gradients that reduce over H),
// rx_gemm_kloop_kernel for longer reductions (multiples of 64, optional split-K), rx_gemm_tn_kernel for the weight
// gradients (both operands transposed: LDS-staged, see there).
//
//     x = hi + lo,  hi = bf16(x), lo = bf16(x - hi);      A.B ~= Ahi.Bhi + Ahi.Blo + Alo.Bhi   (fp32 accumulate)
//
// The dropped lo.lo term and the residual of the split are ~2^-17 relative, so results stay at fp32-level parity (tests
// assert 1e-3 on logits; measured ~1e-5) while v_mfma_f32_32x32x16_bf16 does 16/3 = 5.3x the work per cycle of the
// exact-fp32 MFMA (b4r_gemm.hip: the B4R_GEMM_F32 mode and the path for every other shape).
//
// A workgroup = 4 waves = 128 rows of A; a wave owns 32 rows.  The 32x32x16 operand map (lane l: row = l&31,
// k = 8*(l>>5)+j) makes a fragment 8 consecutive k of one row, so the wave's whole 32 x K strip of A is split ONCE into
// registers and stays there while the workgroup sweeps its share of N, 32 columns per step.
//   B as [K,N] (Keras kernels): a fragment is 8 coalesced dword loads (32 lanes = one 128-byte line each), straight into
//     registers, one step ahead; no LDS, no barriers (rx_gemm_kn_kernel).
//   B as [N,K] (weights used transposed, the tied item table): fragment-shaped loads would touch 32 lines per
//     instruction and saturate the address coalescer (measured: 60 us of a 68 us launch), so the 32 x K tile is fetched
//     ONCE per workgroup with fully coalesced 16-byte loads, split into bf16 hi/lo planes on its way into a
//     double-buffered LDS tile and read back by the 4 waves as ready-made 16-byte fragments; one barrier per step
//     (rx_gemm_nk_kernel).
// In both kernels everything needed by step s+1 is requested before the MFMAs of step s, i.e. it is OLDER than the
// stores of step s, so a wave never waits on its own stores (vmcnt retires in issue order); and there is no
// data-dependent control flow in the loop: out-of-range columns are clamped onto valid ones instead of being guarded,
// because after a control-flow join hipcc falls back to s_waitcnt vmcnt(0) and serialises every round trip (guide,
// "three .s-level traps", (c)).  Hence the shape contract of b4r_gemm_rx_supported().
// Epilogue: accumulators (row = register, column = lane) are transposed through a wave-private LDS tile so that every
// global access (C, the GELU pre-activation copy, the residual) is a 16-byte piece of a full 128-byte line.
#include <stdlib.h>

#include "b4r_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ST_LD = 32;                          // epilogue tile row stride (floats): conflict-free b32 writes / b128 reads
constexpr int STAGE_FLOATS = 4 * 32 * ST_LD;       // 4 waves x 32 rows
constexpr int BROW_MAX = 36;                       // dwords per row of a B plane at K = 64 (128 B of bf16 + 16 B pad)
constexpr size_t KN_LDS_BYTES = STAGE_FLOATS * sizeof(float);
constexpr size_t NK_LDS_BYTES = STAGE_FLOATS * sizeof(float) + 2 * 2 * 32 * BROW_MAX * 4;  // + 2 buffers x (hi, lo)

struct RxP {
  const float* A; const float* B; float* C; const float* bias; float* C2; const float* R;
  int lda, ldb, ldc, ldc2, ldr;
  int M, N, K;
  int n_store;                    // columns that may be written: N rounded up to 4 (<= ldc; pad columns are scratch)
  int n_splits, steps_per_split;  // n-steps (32 columns each) per workgroup
  int n_items;                    // row blocks x column splits (the grid is this rounded up to 8, see xcd_logical_id)
  int k_chunks_per_split;         // K-loop kernel: chunks of 64 per blockIdx.z (all of them when the grid has one z)
  int64_t slab_stride;            // K-loop kernel with several z: partial products go to C + z * slab_stride
  float qscale; int qcols;
  DropArgs drop;
  // B4R_EPI_BIAS_DROP_RES_LN: C2 = LayerNorm(C) * ln_gamma + ln_beta, row statistics of C
  const float* ln_gamma; const float* ln_beta; float* ln_mean; float* ln_rstd; float ln_eps;
  // B4R_EPI_ADD_RES_LN_BWD: the normalisation's input [M,64] (ln_mean / ln_rstd / ln_gamma are inputs here, C2 = partials)
  const float* ln_z; int ln_ldz;
  // ... of the embedding stage (ln_ids != nullptr): z = ln_table[id] + ln_pos[row % ln_L] is recomputed, dy first goes back
  // through the embedding dropout (`drop`, element index row * 64 + col)
  const int64_t* ln_ids; const float* ln_table; const float* ln_pos; int ln_L, ln_V;
  float* C3; int ldc3;   // B4R_EPI_BIAS_GELU_LN: the pre-activation
  // gathered A rows (tile kernel, B4R_EPI_BIAS_GELU_LN only): row m of the product reads row
  // clamp(a_idx[m], 0, a_add_per - 1) + (m / a_per) * a_add_per of A, as b4r_gather_rows does; a_copy [M, a_copy_ld] (optional)
  // receives the gathered rows (the weight-gradient product of the backward pass reads them again)
  const int64_t* a_idx; int64_t a_add_per; int a_per; float* a_copy; int a_copy_ld;
};

__device__ __forceinline__ void split8(const f32x8 x, bf16x8& hi, bf16x8& lo) { b4r_split8(x, hi, lo); }

__device__ __forceinline__ f32x16 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ f32x8 load8_contig(const float* ptr) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(ptr);
  const f32x4 b = *reinterpret_cast<const f32x4*>(ptr + 4);
  return (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__device__ __forceinline__ f32x8 load8_strided(const float* ptr, int ld) {
  f32x8 x;
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = ptr[(int64_t)j * ld];
  return x;
}

// 8 consecutive elements of one row (two hash groups when idx % 4 == 0)
__device__ __forceinline__ f32x8 drop8(const DropCtx& c, const f32x8 x, uint64_t idx) {
  const f32x4 a = b4r_drop4(c, (f32x4){x[0], x[1], x[2], x[3]}, idx);
  const f32x4 b = b4r_drop4(c, (f32x4){x[4], x[5], x[6], x[7]}, idx + 4);
  return (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2).  Workgroups that share an operand (the
// column splits of one row block, the output tiles of one row slice) are given consecutive LOGICAL ids, so the logical
// id is chosen such that a run of consecutive logical ids sits on one XCD: logical = (id % 8) * ceil(n/8) + id / 8.
// Ids whose logical id falls beyond n (n not a multiple of 8) return without work; the grid is rounded up to 8.
// (ML-1M step, same box: 0.922 ms with the mapping, 0.931 ms without (B4R_XCD=0); single kernels move by < 1 us -- the
// re-reads were mostly served by the shared MALL already.)
__device__ __forceinline__ int xcd_logical_id(int id, int n) {
  if (n < 0) return id;   // mapping switched off (B4R_XCD=0): n is passed negated
  const int per = (n + 7) >> 3;
  return (id & 7) * per + (id >> 3);
}
inline bool xcd_on() { static const bool on = !(getenv("B4R_XCD") && atoi(getenv("B4R_XCD")) == 0); return on; }
inline unsigned xcd_grid(int64_t n) { return (unsigned)(((n + 7) >> 3) << 3); }

constexpr int EPI_ADD_RES_LN_BWD_EMBED = 11;   // internal: B4R_EPI_ADD_RES_LN_BWD with ln_ids set
constexpr bool epi_has_bias(int e) {
  return e == B4R_EPI_BIAS || e == B4R_EPI_BIAS_QSCALE || e == B4R_EPI_BIAS_GELU || e == B4R_EPI_BIAS_DROP_RES ||
         e == B4R_EPI_BIAS_TANH || e == B4R_EPI_BIAS_DROP_RES_LN || e == B4R_EPI_BIAS_GELU_LN;
}
constexpr bool epi_has_r(int e) {
  return e == B4R_EPI_BIAS_DROP_RES || e == B4R_EPI_GELU_BWD || e == B4R_EPI_ADD_RES || e == B4R_EPI_BIAS_DROP_RES_LN ||
         e == B4R_EPI_ADD_RES_LN_BWD || e == EPI_ADD_RES_LN_BWD_EMBED;
}

// the wave's 32 x K strip of A, split into hi/lo fragments (row must be valid: M % 32 == 0 and the wave is live)
template <bool A_DROP, int NKB>
__device__ __forceinline__ void load_a_strip(const RxP& p, const DropCtx& dctx, int row, int h, bf16x8 (&ah)[NKB], bf16x8 (&al)[NKB]) {
  const float* arow = p.A + (int64_t)row * p.lda + 8 * h;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    f32x8 x = load8_contig(arow + 16 * kb);
    if (A_DROP) x = drop8(dctx, x, (uint64_t)row * (uint64_t)p.K + (uint64_t)(16 * kb + 8 * h));
    split8(x, ah[kb], al[kb]);
  }
}

// a lane's 4 epilogue columns are either all writable or all beyond n_store (n_store % 4 == 0): the latter are clamped
// onto the last group, which then rewrites identical values -- no divergence anywhere
template <int EPI>
__device__ __forceinline__ f32x4 load_bias4(const RxP& p, int n0, int c4) {
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (epi_has_bias(EPI)) {
    const int colg = min(n0 + c4, p.n_store - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = p.bias[min(colg + e, p.N - 1)];
  }
  return bv;
}

// the lane's 4 x float4 of the residual / pre-activation operand R for the 32 x 32 tile at (m0, n0): requested by the
// callers one step AHEAD (with the B tile), so its HBM latency is not paid inside the epilogue
struct RTile { f32x4 v[4]; };
template <int EPI>
__device__ __forceinline__ RTile load_r_tile(const RxP& p, int m0, int n0, int lane) {
  RTile t;
  if (epi_has_r(EPI)) {
    const int c4 = (lane & 7) * 4, rsub = lane >> 3;
    const int col = min(n0 + c4, p.n_store - 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) t.v[i] = *reinterpret_cast<const f32x4*>(p.R + (int64_t)(m0 + rsub + 8 * i) * p.ldr + col);
  }
  return t;
}

template <int EPI>
__device__ __forceinline__ void epilogue_tile(const RxP& p, const DropCtx& dctx, const f32x16& acc, const f32x4 bv,
                                              const RTile& rt, float* stage, int m0, int n0, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) stage[((reg & 3) + 8 * (reg >> 2) + 4 * h) * ST_LD + r] = acc[reg];
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed; no other wave touches the tile
  const int c4 = (lane & 7) * 4, rsub = lane >> 3;
  const int col = min(n0 + c4, p.n_store - 4);
  f32x4 vin[4];
  const f32x4 (&rr)[4] = rt.v;
#pragma unroll
  for (int i = 0; i < 4; ++i) vin[i] = *reinterpret_cast<const f32x4*>(&stage[(rsub + 8 * i) * ST_LD + (col - n0)]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + rsub + 8 * i;
    f32x4 o, o2 = {0.f, 0.f, 0.f, 0.f}, dz = {0.f, 0.f, 0.f, 0.f};
    if (EPI == B4R_EPI_BIAS_DROP_RES) dz = b4r_drop4(dctx, vin[i] + bv, (uint64_t)row * (uint64_t)p.N + (uint64_t)col);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = vin[i][e];
      float y;
      if (EPI == B4R_EPI_NONE) y = a;
      else if (EPI == B4R_EPI_BIAS) y = a + bv[e];
      else if (EPI == B4R_EPI_BIAS_QSCALE) y = (a + bv[e]) * ((col + e < p.qcols) ? p.qscale : 1.0f);
      else if (EPI == B4R_EPI_BIAS_GELU) { o2[e] = a + bv[e]; y = b4r_gelu_fast(o2[e]); }
      else if (EPI == B4R_EPI_BIAS_DROP_RES) y = rr[i][e] + dz[e];
      else if (EPI == B4R_EPI_GELU_BWD) y = a * b4r_gelu_grad_fast(rr[i][e]);
      else if (EPI == B4R_EPI_ADD_RES) y = a + rr[i][e];
      else y = tanhf(a + bv[e]);
      o[e] = y;
    }
    *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = o;
    if (EPI == B4R_EPI_BIAS_GELU) *reinterpret_cast<f32x4*>(p.C2 + (int64_t)row * p.ldc2 + col) = o2;
  }
}

// B4R_EPI_BIAS_DROP_RES_LN in the 64 x 64 tile kernel with N = 64: the workgroup holds whole rows, wave (wm, wn) the 32 x 32
// quarter at (32 wm, 32 wn).  z = R + dropout(acc + bias) -> C as in BIAS_DROP_RES; then the LayerNorm of b4r_ln_fwd on the
// values still in registers: each wave reduces its 32 columns of a row to (mean, sum of squared deviations) over the 8
// lanes that share the row, the two waves of a row exchange them through their (now idle) staging areas and merge them
// (Chan et al.: M2 = M2a + M2b + (ma - mb)^2 n/2 with n = 32 per half), so the variance is a two-pass one like the stand-alone
// kernel's.  `live` = the quarter's rows exist (M % 32 == 0, so a quarter is whole or absent); every wave takes the barrier.
// GELU: B4R_EPI_BIAS_GELU_LN instead -- z = gelu(acc + bias) (-> C), the pre-activation goes to C3, no residual / dropout
template <bool GELU>
__device__ __forceinline__ void epilogue_tile_ln(const RxP& p, const DropCtx& dctx, const f32x16& acc, const f32x4 bv,
                                                 const RTile& rt, float* stage, float* stage_other, int m0, int n0, int lane,
                                                 bool live) {
  const int r = lane & 31, h = lane >> 5;
  const int c4 = (lane & 7) * 4, rsub = lane >> 3;
  const int col = n0 + c4;
  f32x4 z[4];
  float mean_w[4], m2_w[4];
  if (live) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) stage[((reg & 3) + 8 * (reg >> 2) + 4 * h) * ST_LD + r] = acc[reg];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    f32x4 vin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vin[i] = *reinterpret_cast<const f32x4*>(&stage[(rsub + 8 * i) * ST_LD + c4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + rsub + 8 * i;
      if (GELU) {
        const f32x4 pre = vin[i] + bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) z[i][e] = b4r_gelu_fast(pre[e]);
        *reinterpret_cast<f32x4*>(p.C3 + (int64_t)row * p.ldc3 + col) = pre;
      } else {
        z[i] = rt.v[i] + b4r_drop4(dctx, vin[i] + bv, (uint64_t)row * (uint64_t)p.N + (uint64_t)col);
      }
      *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = z[i];
      float sw = (z[i][0] + z[i][1]) + (z[i][2] + z[i][3]);
      sw += __shfl_xor(sw, 1, 64); sw += __shfl_xor(sw, 2, 64); sw += __shfl_xor(sw, 4, 64);
      mean_w[i] = sw * (1.0f / 32.0f);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float dd = z[i][e] - mean_w[i]; q += dd * dd; }
      q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
      m2_w[i] = q;
    }
    // the tile reads above are complete for the whole wave (the shuffles consumed them): reuse the first 64 floats
    if ((lane & 7) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float2*>(&stage[2 * (rsub + 8 * i)]) = make_float2(mean_w[i], m2_w[i]);
    }
  }
  __syncthreads();
  if (!live) return;
  const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + col);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_beta + col);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + rsub + 8 * i;
    const float2 o = *reinterpret_cast<const float2*>(&stage_other[2 * (rsub + 8 * i)]);
    const float dm = mean_w[i] - o.x;
    const float mean = 0.5f * (mean_w[i] + o.x);
    const float var = (m2_w[i] + o.y + dm * dm * 16.0f) * (1.0f / 64.0f);
    const float rstd = rsqrtf(var + p.ln_eps);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float inv = rstd * g[e];
      y[e] = z[i][e] * inv + (b[e] - mean * inv);
    }
    *reinterpret_cast<f32x4*>(p.C2 + (int64_t)row * p.ldc2 + col) = y;
    if (n0 == 0 && (lane & 7) == 0) {
      if (p.ln_mean) p.ln_mean[row] = mean;
      if (p.ln_rstd) p.ln_rstd[row] = rstd;
    }
  }
}

// B4R_EPI_ADD_RES_LN_BWD, same geometry: dy = acc + R never leaves the registers; the LayerNorm backward of b4r_ln_bwd
//   g = dy gamma ; xhat = (z - mean) rstd ; dz = rstd (g - mean_c(g) - xhat mean_c(g xhat))  -> C
// needs two row sums over the 64 columns (8-lane shuffles, then one exchange between the two waves of a row) and yields the
// column sums of dy xhat / dy over the tile's 64 rows (shuffles over the 8 row groups of a wave, then wave (0, wn) adds wave
// (1, wn)'s): partial[lid][2][64], summed over the workgroups in slab order by the caller's deferred reduction.  Both
// exchanges go through the waves' idle staging areas and share ONE barrier (the column sums do not depend on the row sums).
template <bool EMBED>
__device__ __forceinline__ void epilogue_tile_ln_bwd(const RxP& p, const DropCtx& dctx, const f32x16& acc, const RTile& rt, float* stage,
                                                     const float* stage_row_other, const float* stage_col_other, int lid,
                                                     int m0, int n0, int lane, int wm, bool live) {
  const int r = lane & 31, h = lane >> 5;
  const int c4 = (lane & 7) * 4, rsub = lane >> 3;
  const int col = n0 + c4;
  f32x4 g[4], xh[4];
  float rstd[4], s1w[4], s2w[4];
  f32x4 dgp = {0.f, 0.f, 0.f, 0.f}, dbp = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    f32x4 zz[4];
    float mean[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + rsub + 8 * i;
      if (EMBED) {
        int64_t id = p.ln_ids[row];
        if (id < 0 || id >= p.ln_V) id = 0;   // as the forward: out-of-range ids read the PAD row
        zz[i] = *reinterpret_cast<const f32x4*>(p.ln_table + id * 64 + col) +
                *reinterpret_cast<const f32x4*>(p.ln_pos + (int64_t)(row % p.ln_L) * 64 + col);
      } else {
        zz[i] = *reinterpret_cast<const f32x4*>(p.ln_z + (int64_t)row * p.ln_ldz + col);
      }
      mean[i] = p.ln_mean[row];
      rstd[i] = p.ln_rstd[row];
    }
    const f32x4 gam = *reinterpret_cast<const f32x4*>(p.ln_gamma + col);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) stage[((reg & 3) + 8 * (reg >> 2) + 4 * h) * ST_LD + r] = acc[reg];
    __builtin_amdgcn_s_waitcnt(0xC07F);
    f32x4 vin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vin[i] = *reinterpret_cast<const f32x4*>(&stage[(rsub + 8 * i) * ST_LD + c4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 dy = vin[i] + rt.v[i];
      if (EMBED) dy = b4r_drop4(dctx, dy, (uint64_t)(m0 + rsub + 8 * i) * 64u + (uint64_t)col);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xe = (zz[i][e] - mean[i]) * rstd[i];
        const float ge = dy[e] * gam[e];
        xh[i][e] = xe; g[i][e] = ge;
        s1 += ge; s2 += ge * xe;
        dgp[e] += dy[e] * xe;
        dbp[e] += dy[e];
      }
      s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64); s1 += __shfl_xor(s1, 4, 64);
      s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64); s2 += __shfl_xor(s2, 4, 64);
      s1w[i] = s1; s2w[i] = s2;
    }
#pragma unroll
    for (int o = 8; o <= 32; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dgp[e] += __shfl_xor(dgp[e], o, 64);
        dbp[e] += __shfl_xor(dbp[e], o, 64);
      }
    }
    // the tile has been read by the whole wave (the shuffles above consumed it): floats [0,64) row sums, [64,128) column sums
    if ((lane & 7) == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<float2*>(&stage[2 * (rsub + 8 * i)]) = make_float2(s1w[i], s2w[i]);
    }
    if (rsub == 0) {
      *reinterpret_cast<f32x4*>(&stage[64 + c4]) = dgp;
      *reinterpret_cast<f32x4*>(&stage[96 + c4]) = dbp;
    }
  } else {
    stage[64 + lane] = 0.f;   // an absent lower quarter adds nothing to the column sums
  }
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + rsub + 8 * i;
    const float2 o = *reinterpret_cast<const float2*>(&stage_row_other[2 * (rsub + 8 * i)]);
    const float c1 = (s1w[i] + o.x) * (1.0f / 64.0f), c2 = (s2w[i] + o.y) * (1.0f / 64.0f);
    f32x4 dz;
#pragma unroll
    for (int e = 0; e < 4; ++e) dz[e] = rstd[i] * (g[i][e] - c1 - xh[i][e] * c2);
    *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = dz;
  }
  if (wm == 0 && rsub == 0) {
    const f32x4 og = *reinterpret_cast<const f32x4*>(&stage_col_other[64 + c4]);
    const f32x4 ob = *reinterpret_cast<const f32x4*>(&stage_col_other[96 + c4]);
    float* dst = p.C2 + (int64_t)lid * 128 + col;
    *reinterpret_cast<f32x4*>(dst) = dgp + og;
    *reinterpret_cast<f32x4*>(dst + 64) = dbp + ob;
  }
}

// contract (b4r_gemm_rx_supported): K = 16*NKB; M % 32 == 0; all operands 16-byte aligned with ld % 4 == 0;
// columns [N, n_store) of C (and C2) may be written, of R may be read

// ---- B as [K,N]: register operands, no barriers ----------------------------------------------------------------------
template <int EPI, bool A_DROP, int NKB>
__global__ __launch_bounds__(256) void rx_gemm_kn_kernel(RxP p) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int lid = xcd_logical_id(blockIdx.x, p.n_items);
  if (lid >= abs(p.n_items)) return;   // block-uniform, before any barrier
  const int mblock = lid / p.n_splits, split = lid % p.n_splits;
  const int m0 = mblock * 128 + wave * 32;
  if (m0 >= p.M) return;  // wave-uniform; no barriers in this kernel
  const int total_steps = (p.N + 31) / 32;
  const int s_begin = split * p.steps_per_split;
  const int s_end = min(total_steps, s_begin + p.steps_per_split);
  if (s_begin >= s_end) return;
  DropCtx dctx = b4r_drop_ctx(p.drop);
  float* stage = s_lds + wave * (32 * ST_LD);
  const int c4 = (lane & 7) * 4;

  bf16x8 ah[NKB], al[NKB];
  load_a_strip<A_DROP, NKB>(p, dctx, m0 + r, h, ah, al);

  auto load_b_raw = [&](int n0, int kb) -> f32x8 {
    const int col = min(n0 + r, p.N - 1);
    return load8_strided(p.B + (int64_t)(16 * kb + 8 * h) * p.ldb + col, p.ldb);
  };
  f32x8 braw[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) braw[kb] = load_b_raw(s_begin * 32, kb);
  f32x4 bias_next = load_bias4<EPI>(p, s_begin * 32, c4);
  RTile r_next = load_r_tile<EPI>(p, m0, s_begin * 32, lane);

  for (int s = s_begin; s < s_end; ++s) {
    const int n0 = s * 32;
    bf16x8 bh[NKB], bl[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) split8(braw[kb], bh[kb], bl[kb]);
    const f32x4 bv = bias_next;
    const RTile rt = r_next;
    // unconditional look-ahead (the last step re-requests its own tile: cheaper than a branch, see header)
    const int n_next = min(n0 + 32, (s_end - 1) * 32);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) braw[kb] = load_b_raw(n_next, kb);
    bias_next = load_bias4<EPI>(p, n_next, c4);
    r_next = load_r_tile<EPI>(p, m0, n_next, lane);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = mfma3(ah[kb], al[kb], bh[kb], bl[kb], acc);
    epilogue_tile<EPI>(p, dctx, acc, bv, rt, stage, m0, n0, lane);
  }
}

// ---- B as [N,K]: workgroup-shared, double-buffered bf16 hi/lo tile in LDS --------------------------------------------
template <int EPI, bool A_DROP, int NKB>
__global__ __launch_bounds__(256) void rx_gemm_nk_kernel(RxP p) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  constexpr int F4_PER_ROW = 4 * NKB;                 // float4 per tile row (K / 4)
  constexpr int NLD = (32 * F4_PER_ROW) / 256;        // float4 per thread per tile: 2 (K = 64) or 1 (K = 32)
  constexpr int BROW = 8 * NKB + 4;                   // dwords per plane row: K bf16 + 16 bytes pad (conflict-free b128)
  constexpr int PLANE = 32 * BROW * 4;                // bytes per plane
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int lid = xcd_logical_id(blockIdx.x, p.n_items);
  if (lid >= abs(p.n_items)) return;   // block-uniform, before any barrier
  const int mblock = lid / p.n_splits, split = lid % p.n_splits;
  const int m0 = mblock * 128 + wave * 32;
  const bool live = m0 < p.M;                         // wave-uniform; dead waves still take part in the barriers
  const int total_steps = (p.N + 31) / 32;
  const int s_begin = split * p.steps_per_split;
  const int s_end = min(total_steps, s_begin + p.steps_per_split);  // block-uniform
  DropCtx dctx = b4r_drop_ctx(p.drop);
  float* stage = s_lds + wave * (32 * ST_LD);
  char* bbuf = reinterpret_cast<char*>(s_lds + STAGE_FLOATS);      // [2 buffers][hi, lo][32][BROW dwords]
  const int c4 = (lane & 7) * 4;

  bf16x8 ah[NKB], al[NKB];
  load_a_strip<A_DROP, NKB>(p, dctx, min(m0 + r, p.M - 1), h, ah, al);

  f32x4 raw[NLD];
  auto fetch_b = [&](int n0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = i * 256 + tid;
      const int trow = f / F4_PER_ROW, tc4 = (f % F4_PER_ROW) * 4;
      raw[i] = *reinterpret_cast<const f32x4*>(p.B + (int64_t)min(n0 + trow, p.N - 1) * p.ldb + tc4);
    }
  };
  auto stash_b = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int f = i * 256 + tid;
      const int trow = f / F4_PER_ROW, tc4 = (f % F4_PER_ROW) * 4;
      bf16x4 hi, lo;
      b4r_split4(raw[i], hi, lo);
      char* dst = bbuf + buf * (2 * PLANE) + trow * (BROW * 4) + tc4 * 2;
      *reinterpret_cast<bf16x4*>(dst) = hi;
      *reinterpret_cast<bf16x4*>(dst + PLANE) = lo;
    }
  };

  if (s_begin < s_end) {
    fetch_b(s_begin * 32);
    stash_b(0);
  }
  f32x4 bv = load_bias4<EPI>(p, s_begin * 32, c4);
  const int mr = min(m0, p.M - 32);                    // dead waves (m0 >= M) still issue the look-ahead: keep it in range
  RTile rt = load_r_tile<EPI>(p, mr, s_begin * 32, lane);
  __syncthreads();

  for (int s = s_begin; s < s_end; ++s) {
    const int n0 = s * 32, cur = (s - s_begin) & 1;
    const int n_next = min(n0 + 32, (s_end - 1) * 32);
    fetch_b(n_next);                                   // in flight under the MFMAs and older than this step's stores
    const f32x4 bias_next = load_bias4<EPI>(p, n_next, c4);
    const RTile r_next = load_r_tile<EPI>(p, mr, n_next, lane);
    bf16x8 bh[NKB], bl[NKB];
    const char* src = bbuf + cur * (2 * PLANE) + r * (BROW * 4) + 16 * h;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      bh[kb] = *reinterpret_cast<const bf16x8*>(src + 32 * kb);
      bl[kb] = *reinterpret_cast<const bf16x8*>(src + PLANE + 32 * kb);
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = mfma3(ah[kb], al[kb], bh[kb], bl[kb], acc);
    if (live) epilogue_tile<EPI>(p, dctx, acc, bv, rt, stage, m0, n0, lane);
    stash_b(cur ^ 1);                                  // the other buffer: nobody reads it during this step
    bv = bias_next;
    rt = r_next;
    __syncthreads();
  }
}

// ---- K > 64 (multiple of 64), N <= 32*NT: the K loop -----------------------------------------------------------------
// Used by the products whose reduction dimension is the FFN inner size or 3H (FFN-out, d(FFN-in input), d(QKV input)): their
// N is the hidden size, so a wave keeps all NT column tiles of its 32 rows in accumulators and streams K in chunks of 64:
// A chunk = 8 x 16-byte fragment loads per lane (requested one chunk ahead); B chunk = NT tiles, either register
// fragments of coalesced dword loads ([K,N]) or a workgroup-shared double-buffered bf16 hi/lo LDS tile ([N,K]).
template <bool B_NK, int EPI, bool A_DROP, int NT>
__global__ __launch_bounds__(256) void rx_gemm_kloop_kernel(RxP p) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  constexpr int BROW = 36, PLANE = NT * 32 * BROW * 4;     // B planes: [NT*32 rows][64 bf16 + pad]
  constexpr int NLD = (NT * 32 * 16) / 256;                // float4 per thread per B chunk tile ([N,K] layout)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const int nb = blockIdx.y * (32 * NT);            // first column of this workgroup's pass over N
  const bool live = m0 < p.M;
  const int c_begin = blockIdx.z * p.k_chunks_per_split;
  const int nchunks = min(p.K / 64, c_begin + p.k_chunks_per_split);   // end of this workgroup's chunk range
  DropCtx dctx = b4r_drop_ctx(p.drop);
  float* stage = s_lds + wave * (32 * ST_LD);
  char* bbuf = reinterpret_cast<char*>(s_lds + STAGE_FLOATS);
  const int c4 = (lane & 7) * 4;
  const int arow = min(m0 + r, p.M - 1);

  f32x8 araw[4];
  auto fetch_a = [&](int c) {
    const float* ap = p.A + (int64_t)arow * p.lda + 64 * c + 8 * h;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) araw[kb] = load8_contig(ap + 16 * kb);
  };
  f32x4 braw[B_NK ? NLD : 1];
  auto fetch_b = [&](int c) {
    if constexpr (B_NK) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int f = i * 256 + tid;
        const int trow = f >> 4, tc4 = (f & 15) * 4;
        braw[i] = *reinterpret_cast<const f32x4*>(p.B + (int64_t)min(nb + trow, p.N - 1) * p.ldb + 64 * c + tc4);
      }
    }
  };
  auto stash_b = [&](int buf) {
    if constexpr (B_NK) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int f = i * 256 + tid;
        const int trow = f >> 4, tc4 = (f & 15) * 4;
        bf16x4 hi, lo;
        b4r_split4(braw[i], hi, lo);
        char* dst = bbuf + buf * (2 * PLANE) + trow * (BROW * 4) + tc4 * 2;
        *reinterpret_cast<bf16x4*>(dst) = hi;
        *reinterpret_cast<bf16x4*>(dst + PLANE) = lo;
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  // the residual tiles the epilogue will add are requested now: they arrive during the K loop
  RTile rt[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) rt[j] = load_r_tile<EPI>(p, min(m0, p.M - 32), min(nb + 32 * j, p.N - 1), lane);

  fetch_a(c_begin);
  fetch_b(c_begin);
  stash_b(c_begin & 1);
  if (B_NK) __syncthreads();

  for (int c = c_begin; c < nchunks; ++c) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x8 x = araw[kb];
      if (A_DROP) x = drop8(dctx, x, (uint64_t)arow * (uint64_t)p.K + (uint64_t)(64 * c + 16 * kb + 8 * h));
      split8(x, ah[kb], al[kb]);
    }
    const int cn = min(c + 1, nchunks - 1);   // unconditional look-ahead (the last chunk re-requests itself)
    fetch_a(cn);
    fetch_b(cn);
    const int cur = c & 1;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        bf16x8 bh, bl;
        if constexpr (B_NK) {
          const char* src = bbuf + cur * (2 * PLANE) + (32 * j + r) * (BROW * 4) + 16 * h + 32 * kb;
          bh = *reinterpret_cast<const bf16x8*>(src);
          bl = *reinterpret_cast<const bf16x8*>(src + PLANE);
        } else {
          const int col = min(nb + 32 * j + r, p.N - 1);
          split8(load8_strided(p.B + (int64_t)(64 * c + 16 * kb + 8 * h) * p.ldb + col, p.ldb), bh, bl);
        }
        acc[j] = mfma3(ah[kb], al[kb], bh, bl, acc[j]);
      }
    }
    if (B_NK) {
      stash_b(cur ^ 1);
      __syncthreads();
    }
  }
  if (live) {
    RxP q = p;
    q.C = p.C + (int64_t)blockIdx.z * p.slab_stride;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (nb + 32 * j < p.N)
        epilogue_tile<EPI>(q, dctx, acc[j], load_bias4<EPI>(p, nb + 32 * j, c4), rt[j], stage, m0, nb + 32 * j, lane);
    }
  }
}

// ---- wide shapes (K >= 128 and N >= 128: every dense layer of the 128- and 256-wide configurations) ------------------
// A classic LDS-tiled product: a workgroup owns a 128 x 128 output tile (4 waves as 2 x 2, 64 x 64 each = 2 x 2
// accumulator tiles), K advances in chunks of 32.  Both operand chunks are fetched with full-line 16-byte loads (one chunk
// ahead, in registers), converted ONCE per workgroup to bf16 hi / lo and stored as LDS images; fragments are 16-byte row
// reads (A, and B when it is [N,K]) or transposed reads (B as [K,N]).  The K-loop kernel above re-reads its A strip from
// L2 once per 128-column pass with fragment-shaped loads, which left the ML-20M shapes at ~15 % of the matrix-core peak.
// A / [N,K]-B image: [16-row tile][hi, lo][16 rows x 64 B], 16-byte chunk index XORed with (-(row>>2))&3 (conflict-free
// for the 32x32x16 row fragment: checked as in b4r_rx_tiles.h); [K,N]-B image: the weight-gradient kernel's layout.
// [k rows][64 columns] bf16 image with 128-byte rows (weight-gradient kernel and the [K,N] operand here): the 32-byte
// block index of a row is XORed with 2*bit1(row), see rx_gemm_tn_kernel
__device__ __forceinline__ int tn_img_off(int row, int c4) {   // bytes; c4 = float4 index (0..15) within the 64 columns
  return row * 128 + 32 * ((c4 >> 2) ^ (row & 2)) + 8 * (c4 & 3);
}

constexpr int WIDE_KC = 32;
// WIDE_DEPTH = chunks requested ahead of the one in the matrix cores (template parameter of the kernel).  MEASURED: at the
// ML-1M shapes (K <= 256) two are 1-3 % faster per product when the kernel runs back to back on its own (operands warm in the
// 256 MB last-level cache) but 1.3 % SLOWER over the train step (0.859 vs 0.847 ms), where every product reads what the kernel
// before it just wrote; at the ML-20M shapes (K up to 1024: 32 chunks) two win 1.9 % of the step (5.76 -> 5.66 ms).  So: two for
// the 128 x 128 tiles with K >= B4R_WIDE_DEPTH2_K (256: 5.89 -> 5.79 ms on another box; 512: 5.84), one otherwise (no ML-1M
// product has such a shape).
inline int wide_depth2_k() {
  static const int k = getenv("B4R_WIDE_DEPTH2_K") ? atoi(getenv("B4R_WIDE_DEPTH2_K")) : 256;
  return k;
}
constexpr int wide_lds(int TM, int TN) { return (TM + TN) * 64 * 2 + STAGE_FLOATS * 4; }   // hi + lo images of TM + TN rows of 32 bf16

__device__ __forceinline__ int wide_off(int row, int ch) {   // image 0 (hi); lo is 1024 bytes further
  return (row >> 4) * 2048 + (row & 15) * 64 + 16 * (ch ^ ((0 - (row >> 2)) & 3));
}

// TM x TN = 128 x 128 (wide N) or 64 x 64 (N = 64 with a long K: the products that reduce over the FFN width or 3H);
// the 4 waves always form a 2 x 2 grid of (TM/2) x (TN/2) quarters
template <bool B_NK, int EPI, bool A_DROP, int TM, int TN, int WIDE_DEPTH = 1>
__global__ __launch_bounds__(256, 2) void rx_gemm_wide_kernel(RxP p) {   // two waves per SIMD: without the bound the 128 x 128, depth-2, [K,N] instances take 268 registers = ONE workgroup per CU
  extern __shared__ __attribute__((aligned(16))) char s_w[];
  constexpr int RB = TM / 64, CB = TN / 64;                 // 32-row / 32-column blocks per wave
  constexpr int A_BYTES = TM * 64 * 2, B_BYTES = TN * 64 * 2;
  constexpr int NA = TM * 8 / 256, NB = TN * 8 / 256;       // 16-byte pieces per thread and chunk
  constexpr int KN_HALF = 32 * 128 * 2;                     // [K,N] image of one 64-column half: hi 4 KB | lo 4 KB
  char* sA = s_w;
  char* sB = s_w + A_BYTES;
  float* stage = reinterpret_cast<float*>(s_w + A_BYTES + B_BYTES) + (threadIdx.x >> 6) * (32 * ST_LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + TN - 1) / TN;
  const int lid = xcd_logical_id(blockIdx.x, p.n_items);   // the column tiles of one row block share its A chunks
  if (lid >= abs(p.n_items)) return;
  const int m0 = (lid / ntn) * TM, n0 = (lid % ntn) * TN;
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  const int nchunks = p.K / WIDE_KC;

  // staging maps.  A (and [N,K] B): piece f -> row f>>3, k = 4*(f&7).  [K,N] B: piece f -> k row f / (TN/4), n = 4*(f % (TN/4)).
  struct Regs { f32x4 ra[NA], rb[NB]; };
  auto fetch = [&](Regs& q, int c) {
    const int k0 = c * WIDE_KC;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int f = tid + 256 * it;
      const int m = min(m0 + (f >> 3), p.M - 1);
      int64_t src = m;
      if (p.a_idx) {   // block-uniform
        int64_t pos = p.a_idx[m];
        pos = pos < 0 ? 0 : (pos >= p.a_add_per ? p.a_add_per - 1 : pos);
        src = pos + (int64_t)(m / p.a_per) * p.a_add_per;
      }
      q.ra[it] = *reinterpret_cast<const f32x4*>(p.A + src * p.lda + k0 + 4 * (f & 7));
      if (p.a_copy && n0 == 0 && m0 + (f >> 3) < p.M)
        *reinterpret_cast<f32x4*>(p.a_copy + (int64_t)m * p.a_copy_ld + k0 + 4 * (f & 7)) = q.ra[it];
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int f = tid + 256 * it;
      if (B_NK) {
        q.rb[it] = *reinterpret_cast<const f32x4*>(p.B + (int64_t)min(n0 + (f >> 3), p.N - 1) * p.ldb + k0 + 4 * (f & 7));
      } else {
        const int krow = f / (TN / 4), c4n = f % (TN / 4);
        q.rb[it] = *reinterpret_cast<const f32x4*>(p.B + (int64_t)(k0 + krow) * p.ldb + min(n0 + 4 * c4n, p.n_store - 4));
      }
    }
  };
  auto stash = [&](const Regs& q, int c) {
    const int k0 = c * WIDE_KC;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int f = tid + 256 * it;
      const int row = f >> 3, c4 = f & 7;
      f32x4 va = q.ra[it];
      if (A_DROP) va = b4r_drop4(dctx, va, (uint64_t)min(m0 + row, p.M - 1) * (uint64_t)p.K + (uint64_t)(k0 + 4 * c4));
      bf16x4 hi, lo;
      b4r_split4(va, hi, lo);
      char* da = sA + wide_off(row, c4 >> 1) + 8 * (c4 & 1);
      *reinterpret_cast<bf16x4*>(da) = hi;
      *reinterpret_cast<bf16x4*>(da + 1024) = lo;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int f = tid + 256 * it;
      bf16x4 hi, lo;
      b4r_split4(q.rb[it], hi, lo);
      if (B_NK) {
        const int row = f >> 3, c4 = f & 7;
        char* db = sB + wide_off(row, c4 >> 1) + 8 * (c4 & 1);
        *reinterpret_cast<bf16x4*>(db) = hi;
        *reinterpret_cast<bf16x4*>(db + 1024) = lo;
      } else {
        const int krow = f / (TN / 4), c4n = f % (TN / 4);
        char* db = sB + (c4n >> 4) * KN_HALF + tn_img_off(krow, c4n & 15);
        *reinterpret_cast<bf16x4*>(db) = hi;
        *reinterpret_cast<bf16x4*>(db + 4096) = lo;
      }
    }
  };
  // fragment addresses: kb = 16-wide k block of the chunk (0, 1)
  int a_addr[RB][2], b_addr[CB][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int blk = 0; blk < RB; ++blk) a_addr[blk][kb] = wide_off((TM / 2) * wm + 32 * blk + r, 2 * kb + h);
#pragma unroll
    for (int blk = 0; blk < CB; ++blk) b_addr[blk][kb] = wide_off((TN / 2) * wn + 32 * blk + r, 2 * kb + h);
  }
  const int qq = (lane & 15) >> 2, pp = lane & 3, gb = (lane >> 4) & 1;
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  auto b_frag = [&](int cb, int kb, int plane) -> bf16x8 {
    if (B_NK) return *reinterpret_cast<const bf16x8*>(sB + b_addr[cb][kb] + 1024 * plane);
    // the wave's columns (TN/2)*wn + 32*cb ..: 64-column half and 16-column block inside it
    const int col0 = (TN / 2) * wn + 32 * cb;
    const char* src = sB + (col0 >> 6) * KN_HALF + plane * 4096 +
                      tn_img_off(8 * h + qq, 4 * (((col0 & 63) >> 4) + gb) + pp) + kb * (16 * 128);
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + 4 * 128));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x16 acc[RB][CB];
#pragma unroll
  for (int a = 0; a < RB; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  // WIDE_DEPTH chunks travel ahead of the one being multiplied (with 2 the register sets q0 / q1 alternate)
  auto step = [&](Regs& q, int c) {
    __syncthreads();                       // the previous chunk's images have been consumed
    stash(q, c);
    __syncthreads();
    fetch(q, min(c + WIDE_DEPTH, nchunks - 1));   // unconditional look-ahead (the last chunks are re-requested, never used)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 ah[RB], al[RB], bh[CB], bl[CB];
#pragma unroll
      for (int blk = 0; blk < RB; ++blk) {
        ah[blk] = *reinterpret_cast<const bf16x8*>(sA + a_addr[blk][kb]);
        al[blk] = *reinterpret_cast<const bf16x8*>(sA + a_addr[blk][kb] + 1024);
      }
#pragma unroll
      for (int blk = 0; blk < CB; ++blk) {
        bh[blk] = b_frag(blk, kb, 0);
        bl[blk] = b_frag(blk, kb, 1);
      }
#pragma unroll
      for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) acc[a][b] = mfma3(ah[a], al[a], bh[b], bl[b], acc[a][b]);
    }
  };
  Regs q0, q1;
  fetch(q0, 0);
  if (WIDE_DEPTH == 2) {
    fetch(q1, min(1, nchunks - 1));
    int c = 0;
    for (; c + 1 < nchunks; c += 2) { step(q0, c); step(q1, c + 1); }
    if (c < nchunks) step(q0, c);
  } else {
    for (int c = 0; c < nchunks; ++c) step(q0, c);
  }
  const int c4 = (lane & 7) * 4;
  if constexpr (EPI == B4R_EPI_BIAS_DROP_RES_LN || EPI == B4R_EPI_BIAS_GELU_LN) {
    static_assert(TM == 64 && TN == 64, "the LayerNorm epilogue needs whole rows in one workgroup");
    const int ms = m0 + 32 * wm, ns = 32 * wn;   // N == 64: n0 == 0
    const bool live = ms < p.M;
    RTile rt;
    if (live) rt = load_r_tile<EPI>(p, ms, ns, lane);
    float* stage_other = reinterpret_cast<float*>(s_w + A_BYTES + B_BYTES) + (wave ^ 1) * (32 * ST_LD);
    epilogue_tile_ln<EPI == B4R_EPI_BIAS_GELU_LN>(p, dctx, acc[0][0], load_bias4<EPI>(p, ns, c4), rt, stage, stage_other, ms, ns,
                                                  lane, live);
  } else if constexpr (EPI == B4R_EPI_ADD_RES_LN_BWD || EPI == EPI_ADD_RES_LN_BWD_EMBED) {
    static_assert(TM == 64 && TN == 64, "the LayerNorm epilogue needs whole rows in one workgroup");
    const int ms = m0 + 32 * wm, ns = 32 * wn;
    const bool live = ms < p.M;
    RTile rt;
    if (live) rt = load_r_tile<EPI>(p, ms, ns, lane);
    const float* stage0 = reinterpret_cast<const float*>(s_w + A_BYTES + B_BYTES);
    epilogue_tile_ln_bwd<EPI == EPI_ADD_RES_LN_BWD_EMBED>(p, dctx, acc[0][0], rt, stage, stage0 + (wave ^ 1) * (32 * ST_LD),
                                                          stage0 + (wave ^ 2) * (32 * ST_LD), lid, ms, ns, lane, wm, live);
  } else {
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
      for (int b = 0; b < CB; ++b) {
        const int ms = m0 + (TM / 2) * wm + 32 * a, ns = n0 + (TN / 2) * wn + 32 * b;
        if (ms < p.M && ns < p.N)
          epilogue_tile<EPI>(p, dctx, acc[a][b], load_bias4<EPI>(p, ns, c4), load_r_tile<EPI>(p, ms, ns, lane), stage, ms, ns, lane);
      }
  }
}

// 0: K-loop / register-operand kernels; 128: 128 x 128 tiles; 64: 64 x 64 tiles.  Measured on one box (ML-1M shapes, us):
//   K >= 128: tiles always (FFN-out 22 -> 17.5, dX1 19 -> 16.5, dX 15 -> 12.5; ML-20M step 7.3 -> 5.9 ms)
//   K = 64:   QKV 19.2 -> 16.3, attention-out 12.5 -> 8.9, FFN-in 28.6 -> 26, dctx 10.2 -> 8.1 with tiles; the two products
//             with B as [N,K] and many columns stay on rx_gemm_nk_kernel (GELU' product 30 vs 31.5; vocabulary
//             projection 32 vs 38.5: its one-tile-per-workgroup sweep with the A strip in registers writes faster)
// B4R_WIDE bits switch the three groups off for experiments: 1 = 128-tiles, 2 = 64-tiles, 4 = K = 64 shapes.
template <bool B_NK>
inline int wide_tile(const RxP& p) {
  static const int mode = getenv("B4R_WIDE") ? atoi(getenv("B4R_WIDE")) : 7;
  if (p.K % WIDE_KC != 0 || p.slab_stride != 0) return 0;
  if (p.K < 128) {
    if (!(mode & 4) || p.K != 64 || p.N > 256 || (B_NK && p.N > 128)) return 0;
  }
  if (p.N >= 128) return (mode & 1) ? 128 : 0;
  if (p.N == 64) return (mode & 2) ? 64 : 0;
  return 0;
}

template <bool B_NK, int EPI, bool A_DROP, int T>
void launch_wide(RxP p, hipStream_t s) {
  p.n_items = b4r_cdiv(p.M, T) * b4r_cdiv(p.N, T);
  const dim3 grid(xcd_grid(p.n_items));
  if (!xcd_on()) p.n_items = -p.n_items;
  if constexpr (T == 128) {
    if (p.K >= wide_depth2_k()) {
      (void)b4r_raise_lds((const void*)rx_gemm_wide_kernel<B_NK, EPI, A_DROP, T, T, 2>, wide_lds(T, T), "gemm");
      hipLaunchKernelGGL((rx_gemm_wide_kernel<B_NK, EPI, A_DROP, T, T, 2>), grid, dim3(256), wide_lds(T, T), s, p);
      return;
    }
  }
  (void)b4r_raise_lds((const void*)rx_gemm_wide_kernel<B_NK, EPI, A_DROP, T, T, 1>, wide_lds(T, T), "gemm");
  hipLaunchKernelGGL((rx_gemm_wide_kernel<B_NK, EPI, A_DROP, T, T, 1>), grid, dim3(256), wide_lds(T, T), s, p);
}

template <bool B_NK, int EPI, bool A_DROP>
void launch_kloop(const RxP& p, hipStream_t s) {
  const int nt = b4r_cdiv(p.N, 32);
  dim3 grid((unsigned)b4r_cdiv(p.M, 128), (unsigned)(nt <= 2 ? 1 : b4r_cdiv(nt, 4)),
            (unsigned)b4r_cdiv(p.K / 64, p.k_chunks_per_split));
  if (nt <= 2) {
    const size_t lds = STAGE_FLOATS * sizeof(float) + (B_NK ? 2 * 2 * 2 * 32 * 36 * 4 : 0);
    hipLaunchKernelGGL((rx_gemm_kloop_kernel<B_NK, EPI, A_DROP, 2>), grid, dim3(256), lds, s, p);
  } else {
    const size_t lds = STAGE_FLOATS * sizeof(float) + (B_NK ? 2 * 2 * 4 * 32 * 36 * 4 : 0);
    (void)b4r_raise_lds((const void*)rx_gemm_kloop_kernel<B_NK, EPI, A_DROP, 4>, lds, "gemm");
    hipLaunchKernelGGL((rx_gemm_kloop_kernel<B_NK, EPI, A_DROP, 4>), grid, dim3(256), lds, s, p);
  }
}

template <bool B_NK, int EPI, bool A_DROP>
void launch_rx2(const RxP& p, dim3 grid, hipStream_t s) {
  const int wt = wide_tile<B_NK>(p);
  if (wt == 128) { launch_wide<B_NK, EPI, A_DROP, 128>(p, s); return; }
  if (wt == 64) { launch_wide<B_NK, EPI, A_DROP, 64>(p, s); return; }
  if (p.K > 64) { launch_kloop<B_NK, EPI, A_DROP>(p, s); return; }
  if (B_NK) {
    if (p.K == 64) hipLaunchKernelGGL((rx_gemm_nk_kernel<EPI, A_DROP, 4>), grid, dim3(256), NK_LDS_BYTES, s, p);
    else hipLaunchKernelGGL((rx_gemm_nk_kernel<EPI, A_DROP, 2>), grid, dim3(256), NK_LDS_BYTES, s, p);
  } else {
    if (p.K == 64) hipLaunchKernelGGL((rx_gemm_kn_kernel<EPI, A_DROP, 4>), grid, dim3(256), KN_LDS_BYTES, s, p);
    else hipLaunchKernelGGL((rx_gemm_kn_kernel<EPI, A_DROP, 2>), grid, dim3(256), KN_LDS_BYTES, s, p);
  }
}

template <bool B_NK, int EPI>
void launch_rx(const RxP& p, bool a_drop, dim3 grid, hipStream_t s) {
  if (a_drop) launch_rx2<B_NK, EPI, true>(p, grid, s);
  else launch_rx2<B_NK, EPI, false>(p, grid, s);
}

template <bool B_NK>
int dispatch_rx(const RxP& p, int epi, bool a_drop, dim3 grid, hipStream_t s) {
  if (epi == B4R_EPI_BIAS_DROP_RES_LN) {   // b4r_gemm_ln_supported: N == 64, B as [K,N], no operand dropout
    launch_wide<false, B4R_EPI_BIAS_DROP_RES_LN, false, 64>(p, s);
    return B4R_OK;
  }
  if (epi == B4R_EPI_BIAS_GELU_LN) {
    launch_wide<false, B4R_EPI_BIAS_GELU_LN, false, 64>(p, s);
    return B4R_OK;
  }
  if (epi == B4R_EPI_ADD_RES_LN_BWD) {     // N == 64, B as [N,K]
    if (p.ln_ids) launch_wide<true, EPI_ADD_RES_LN_BWD_EMBED, false, 64>(p, s);
    else launch_wide<true, B4R_EPI_ADD_RES_LN_BWD, false, 64>(p, s);
    return B4R_OK;
  }
  switch (epi) {
    case B4R_EPI_NONE: launch_rx<B_NK, B4R_EPI_NONE>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS: launch_rx<B_NK, B4R_EPI_BIAS>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_QSCALE: launch_rx<B_NK, B4R_EPI_BIAS_QSCALE>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_GELU: launch_rx<B_NK, B4R_EPI_BIAS_GELU>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_DROP_RES: launch_rx<B_NK, B4R_EPI_BIAS_DROP_RES>(p, a_drop, grid, s); break;
    case B4R_EPI_GELU_BWD: launch_rx<B_NK, B4R_EPI_GELU_BWD>(p, a_drop, grid, s); break;
    case B4R_EPI_ADD_RES: launch_rx<B_NK, B4R_EPI_ADD_RES>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_TANH: launch_rx<B_NK, B4R_EPI_BIAS_TANH>(p, a_drop, grid, s); break;
    default: b4r_set_error("gemm: unknown epilogue %d", epi); return B4R_E_BADARG;
  }
  return B4R_OK;
}


// ---- weight gradients: out[Mo,No] = A[R,Mo]^T . B[R,No] --------------------------------------------------------------
// The reduction index k is the ROW index of both operands, so both MFMA operands are columns of row-major data.  A
// workgroup (4 waves, one 32x32 quarter each) owns one (64 x 64 output tile, slice of R) item and walks its slice in
// chunks of KS rows: every thread fetches full 16-byte pieces of whole 256-byte row segments (requested one chunk ahead,
// KS*512 bytes in flight per workgroup), converts them once to bf16 hi / lo and writes two [KS][64] LDS images per
// operand; fragments are then hardware-transposed reads (ds_read_b64_tr_b16: a lane gets one column of 4 consecutive
// rows).  The 32-byte block index of an image row is XORed with 2*bit1(row), which spreads the 4 rows of a half-wave's
// transposed read over all 64 banks.  Dropout of B (the masked upstream gradient) and the bias-gradient column sums
// (of B, or of A for the tied-table / output-bias pair) ride on the staged values, whose column is fixed per thread.
// Partial tiles go to slabs that are summed in a fixed order later (bitwise reproducible).
struct RxTnP {
  const float* A; const float* B; float* slab; float* colsum_slab; float* colsum_a_slab;
  int lda, ldb;
  int R, Mo, No;
  int a_lim, b_lim;                // first column that may not be fetched (Mo / No rounded up to 4), <= ld
  int tiles_i, tiles_j, S, chunk;  // chunk: rows per slice
  DropArgs drop;
  // DGRAD (No = 64, Mo a multiple of 64): dX[R, Mo] = dropout(B) . W^T with W [Mo, ldw] the weight whose gradient `out` is; the
  // workgroup of output tile ti forms columns 64 ti .. of dX.  DGRAD == 2: dX is further multiplied by gelu'(G) (G [R, ldg])
  const float* W; float* dX; const float* G; int ldw, lddx, ldg;
};


template <int KS, bool B_DROP, int DGRAD>
__device__ __forceinline__ void rx_gemm_tn_body(const RxTnP& p, const int block) {
  constexpr int NLD = KS / 16;                       // float4 per thread per operand per chunk
  constexpr int PLANE = KS * 128;                    // bytes per image
  extern __shared__ __attribute__((aligned(16))) char s_tn[];   // [A hi | A lo | B hi | B lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_items = p.tiles_i * p.tiles_j * p.S;
  const int item = xcd_logical_id(block, n_items);   // the tiles of one row slice share its chunks of A and B
  if (item >= n_items) return;                            // block-uniform, before any barrier
  const int z = item / (p.tiles_i * p.tiles_j);
  const int t = item % (p.tiles_i * p.tiles_j);
  const int ti = t / p.tiles_j, tj = t % p.tiles_j;
  const int i0 = ti * 64, j0 = tj * 64;
  const int r_begin = min(p.R, z * p.chunk), r_end = min(p.R, r_begin + p.chunk);
  const bool do_cs = p.colsum_slab != nullptr && ti == 0;
  const bool do_csa = p.colsum_a_slab != nullptr && tj == 0;
  const DropCtx dctx = b4r_drop_ctx(p.drop);

  // staging: thread -> (row tid/16 + 16*j, float4 column tid%16); columns beyond the matrix are clamped onto fetchable
  // ones (their products land in output elements that are never stored)
  const int c4 = tid & 15, srow = tid >> 4;
  const int ca = min(i0 + 4 * c4, p.a_lim - 4), cb = min(j0 + 4 * c4, p.b_lim - 4);
  struct Chunk { f32x4 a[NLD], b[NLD]; };
  auto fetch = [&](Chunk& c, int k0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int row = min(k0 + srow + 16 * j, p.R - 1);
      c.a[j] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)row * p.lda + ca);
      c.b[j] = *reinterpret_cast<const f32x4*>(p.B + (int64_t)row * p.ldb + cb);
    }
  };
  f32x4 cs = {0.f, 0.f, 0.f, 0.f}, csa = {0.f, 0.f, 0.f, 0.f};
  auto stash = [&](const Chunk& c, int k0) {
    const f32x4 (&ra)[NLD] = c.a;
    const f32x4 (&rb)[NLD] = c.b;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int lrow = srow + 16 * j, row = k0 + lrow;
      const bool live = row < r_end;
      f32x4 va = ra[j], vb = rb[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = live ? va[e] : 0.f;
        vb[e] = live ? vb[e] : 0.f;
      }
      if (B_DROP) vb = b4r_drop4(dctx, vb, (uint64_t)row * (uint64_t)p.No + (uint64_t)cb);
      cs += vb;
      csa += va;
      const int off = tn_img_off(lrow, c4);
      bf16x4 ah, al, bh, bl;
      b4r_split4(va, ah, al);
      b4r_split4(vb, bh, bl);
      *reinterpret_cast<bf16x4*>(s_tn + off) = ah;
      *reinterpret_cast<bf16x4*>(s_tn + PLANE + off) = al;
      *reinterpret_cast<bf16x4*>(s_tn + 2 * PLANE + off) = bh;
      *reinterpret_cast<bf16x4*>(s_tn + 3 * PLANE + off) = bl;
    }
  };
  // transposed fragment: 16-lane group G = lane>>4 reads the 4 x 16 block at rows 8h + 4s + (0..3) (+16*kb), columns
  // 32*w + 16*(G&1) ..; lane 4q+pp of the group supplies the address of row q, columns 4pp..4pp+3
  const int qq = (lane & 15) >> 2, pp = lane & 3, gb = (lane >> 4) & 1;
  const int tr_a = tn_img_off(8 * h + qq, 4 * (2 * wm + gb) + pp);
  const int tr_b = tn_img_off(8 * h + qq, 4 * (2 * wn + gb) + pp);
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  auto tr8 = [&](int plane, int addr, int kb) {
    const char* src = s_tn + plane * PLANE + addr + kb * (16 * 128);
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + 4 * 128));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // DGRAD: the input gradient of the same layer from the same staged chunk.  The (dropped) B image is row-major, so its rows
  // are MFMA A fragments by plain 16-byte reads (8 consecutive k of a row sit in one 32-byte block of the image); the B
  // operand W[j][k] is held as register fragments for the wave's 32 columns j.  Quarter (wm, wn) = rows 32 wm.., columns 32 wn..
  bf16x8 wh[KS / 16], wl[KS / 16];
  if (DGRAD) {
#pragma unroll
    for (int kb = 0; kb < KS / 16; ++kb)
      split8(load8_contig(p.W + (int64_t)(i0 + 32 * wn + r) * p.ldw + 16 * kb + 8 * h), wh[kb], wl[kb]);
  }
  auto dgrad = [&](int k0) {
    f32x16 dx, gp;
#pragma unroll
    for (int i = 0; i < 16; ++i) dx[i] = 0.f;
    if (DGRAD == 2) {   // the pre-activations travel while the products run (rows clamped, stores guarded)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = min(k0 + 32 * wm + (reg & 3) + 8 * (reg >> 2) + 4 * h, p.R - 1);
        gp[reg] = p.G[(int64_t)row * p.ldg + i0 + 32 * wn + r];
      }
    }
#pragma unroll
    for (int kb = 0; kb < KS / 16; ++kb) {
      const char* src = s_tn + 2 * PLANE + tn_img_off(32 * wm + r, 4 * kb + 2 * h);
      dx = mfma3(*reinterpret_cast<const bf16x8*>(src), *reinterpret_cast<const bf16x8*>(src + PLANE), wh[kb], wl[kb], dx);
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = k0 + 32 * wm + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float v = dx[reg];
      if (DGRAD == 2) v *= b4r_gelu_grad_fast(gp[reg]);
      if (row < r_end) p.dX[(int64_t)row * p.lddx + i0 + 32 * wn + r] = v;
    }
  };
  // Two register sets alternate: the loads of chunk k+1 are issued at the TOP of iteration k (before the barrier, the
  // conversion and the products of chunk k), so they have a whole iteration to arrive; one set of LDS images.
  auto products = [&]() {
#pragma unroll
    for (int kb = 0; kb < KS / 16; ++kb)
      acc = mfma3(tr8(0, tr_a, kb), tr8(1, tr_a, kb), tr8(2, tr_b, kb), tr8(3, tr_b, kb), acc);
  };
  Chunk c0, c1;
  if (r_begin < r_end) fetch(c0, r_begin);
  for (int k0 = r_begin; k0 < r_end; k0 += 2 * KS) {
    fetch(c1, k0 + KS);              // unconditional look-ahead (rows are clamped; stash() zeroes what lies beyond r_end)
    __syncthreads();                 // every wave is done with the previous chunk's images
    stash(c0, k0);
    __syncthreads();
    products();
    if (DGRAD) dgrad(k0);
    if (k0 + KS >= r_end) break;     // block-uniform
    fetch(c0, k0 + 2 * KS);
    __syncthreads();
    stash(c1, k0 + KS);
    __syncthreads();
    products();
    if (DGRAD) dgrad(k0 + KS);
  }

  // the column sums first (two barriers), the tile's 16 KB of slab stores last: behind the stores the barriers would wait for
  // their acknowledgements
  if (do_cs || do_csa) {             // 16 row groups per column quad, summed in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(s_tn);   // [2][16 row groups][64 columns]
    *reinterpret_cast<f32x4*>(red + srow * 64 + 4 * c4) = cs;
    *reinterpret_cast<f32x4*>(red + 1024 + srow * 64 + 4 * c4) = csa;
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += red[which * 1024 + g * 64 + c];
      if (which == 0 && do_cs && j0 + c < p.No) p.colsum_slab[(int64_t)z * p.No + j0 + c] = sum;
      if (which == 1 && do_csa && i0 + c < p.Mo) p.colsum_a_slab[(int64_t)z * p.Mo + i0 + c] = sum;
    }
  }
  float* slab = p.slab + (int64_t)z * p.Mo * p.No;
  const int col = j0 + 32 * wn + r;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = i0 + 32 * wm + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (row < p.Mo && col < p.No) slab[(int64_t)row * p.No + col] = acc[reg];
  }
}

template <int KS, bool B_DROP, int DGRAD>
__global__ __launch_bounds__(256) void rx_gemm_tn_kernel(RxTnP p) {
  rx_gemm_tn_body<KS, B_DROP, DGRAD>(p, (int)blockIdx.x);
}
// two independent weight-gradient products in ONE launch (the workgroups of the second follow those of the first; n0 is a
// multiple of 8, so both keep their XCD mapping): dWo = ctx^T.dropmask(dz1) and dWqkv = x^T.dqkv of an encoder layer are each too
// short (12 / 17 us) to reach the memory system's rate on their own
#ifndef TN_PAIR_OCC
#define TN_PAIR_OCC 3   // waves per SIMD the register allocation aims at (tools/build_variant.sh -DTN_PAIR_OCC=4 to compare)
#endif
template <int KS, bool B_DROP0, bool B_DROP1>
__global__ __launch_bounds__(256, TN_PAIR_OCC) void rx_gemm_tn_pair_kernel(RxTnP p0, RxTnP p1, int n0) {
  if ((int)blockIdx.x < n0) rx_gemm_tn_body<KS, B_DROP0, 0>(p0, (int)blockIdx.x);
  else rx_gemm_tn_body<KS, B_DROP1, 0>(p1, (int)blockIdx.x - n0);
}

// ---- the same product on 128 x 128 output tiles (Mo and No multiples of 128: the weight gradients of the wide configurations) ------
// With 64 x 64 tiles every operand column block is fetched by Mo/64 or No/64 workgroups: at ML-20M (dW1 = x^T.dfpre, 256 x 1024
// over 25 600 rows) 0.84 GB cross the L2s per launch for 0.13 GB of operands, and a workgroup spends a chunk mostly on staging
// (12 MFMAs per wave against 8 float4 converted per thread).  Here a workgroup of 8 waves owns 128 x 128, a wave 64 x 32 (two blocks of 32 x 32): four times
// the matrix work per staged byte, and two waves per SIMD even with one workgroup on the CU (with 4 waves of 64 x 64 a CU held 4).  Chunks of 32 rows; image rows are 256 bytes, the 64-byte quarter index XORed with row & 3, so the
// four rows of a half-wave's transposed read cover all 64 banks.
constexpr int TN128_KS = 32;
__device__ __forceinline__ int tn128_off(int row, int c4) {   // bytes; c4 = float4 index (0..31) within the 128 columns
  return row * 256 + 32 * ((c4 >> 2) ^ (2 * (row & 3))) + 8 * (c4 & 3);
}
template <bool B_DROP>
__global__ __launch_bounds__(512, 2) void rx_gemm_tn128_kernel(RxTnP p) {
  constexpr int KS = TN128_KS, NLD = KS / 16, PLANE = KS * 256;
  extern __shared__ __attribute__((aligned(16))) char s_tn[];   // [A hi | A lo | B hi | B lo]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 2, wn = wave & 3;   // 8 waves: 2 x 4 grid of 64 x 32 pieces (two 32 x 32 blocks each)
  const int n_items = p.tiles_i * p.tiles_j * p.S;
  const int item = xcd_logical_id((int)blockIdx.x, n_items);
  if (item >= n_items) return;                            // block-uniform, before any barrier
  const int z = item / (p.tiles_i * p.tiles_j);
  const int t = item % (p.tiles_i * p.tiles_j);
  const int ti = t / p.tiles_j, tj = t % p.tiles_j;
  const int i0 = ti * 128, j0 = tj * 128;
  const int r_begin = min(p.R, z * p.chunk), r_end = min(p.R, r_begin + p.chunk);
  const bool do_cs = p.colsum_slab != nullptr && ti == 0;
  const bool do_csa = p.colsum_a_slab != nullptr && tj == 0;
  const DropCtx dctx = b4r_drop_ctx(p.drop);

  const int c4 = tid & 31, srow = tid >> 5;   // thread -> (row tid/32 + 16*j, float4 column tid%32)
  const int ca = i0 + 4 * c4, cb = j0 + 4 * c4;
  struct Chunk { f32x4 a[NLD], b[NLD]; };
  auto fetch = [&](Chunk& c, int k0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int row = min(k0 + srow + 16 * j, p.R - 1);
      c.a[j] = *reinterpret_cast<const f32x4*>(p.A + (int64_t)row * p.lda + ca);
      c.b[j] = *reinterpret_cast<const f32x4*>(p.B + (int64_t)row * p.ldb + cb);
    }
  };
  f32x4 cs = {0.f, 0.f, 0.f, 0.f}, csa = {0.f, 0.f, 0.f, 0.f};
  auto stash = [&](const Chunk& c, int k0) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int lrow = srow + 16 * j, row = k0 + lrow;
      const bool live = row < r_end;
      f32x4 va = c.a[j], vb = c.b[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = live ? va[e] : 0.f;
        vb[e] = live ? vb[e] : 0.f;
      }
      if (B_DROP) vb = b4r_drop4(dctx, vb, (uint64_t)row * (uint64_t)p.No + (uint64_t)cb);
      cs += vb;
      csa += va;
      const int off = tn128_off(lrow, c4);
      bf16x4 ah, al, bh, bl;
      b4r_split4(va, ah, al);
      b4r_split4(vb, bh, bl);
      *reinterpret_cast<bf16x4*>(s_tn + off) = ah;
      *reinterpret_cast<bf16x4*>(s_tn + PLANE + off) = al;
      *reinterpret_cast<bf16x4*>(s_tn + 2 * PLANE + off) = bh;
      *reinterpret_cast<bf16x4*>(s_tn + 3 * PLANE + off) = bl;
    }
  };
  // transposed fragments as in rx_gemm_tn_body; A block b of the wave = columns 64 wm + 32 b .. of the A image, B block = 32 wn ..
  const int qq = (lane & 15) >> 2, pp = lane & 3, gb = (lane >> 4) & 1;
  int tr_a[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) tr_a[b] = tn128_off(8 * h + qq, 4 * (2 * (2 * wm + b) + gb) + pp);
  const int tr_b = tn128_off(8 * h + qq, 4 * (2 * wn + gb) + pp);
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  auto tr8 = [&](int plane, int addr, int kb) {
    const char* src = s_tn + plane * PLANE + addr + kb * (16 * 256);
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(src + 4 * 256));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  f32x16 acc[2];
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[bi][i] = 0.f;
  auto products = [&]() {
#pragma unroll
    for (int kb = 0; kb < KS / 16; ++kb) {
      const bf16x8 bh = tr8(2, tr_b, kb), bl = tr8(3, tr_b, kb);
#pragma unroll
      for (int bi = 0; bi < 2; ++bi) acc[bi] = mfma3(tr8(0, tr_a[bi], kb), tr8(1, tr_a[bi], kb), bh, bl, acc[bi]);
    }
  };
  Chunk c0, c1;
  if (r_begin < r_end) fetch(c0, r_begin);
  for (int k0 = r_begin; k0 < r_end; k0 += 2 * KS) {
    fetch(c1, k0 + KS);              // unconditional look-ahead (rows are clamped; stash() zeroes what lies beyond r_end)
    __syncthreads();                 // every wave is done with the previous chunk's images
    stash(c0, k0);
    __syncthreads();
    products();
    if (k0 + KS >= r_end) break;     // block-uniform
    fetch(c0, k0 + 2 * KS);
    __syncthreads();
    stash(c1, k0 + KS);
    __syncthreads();
    products();
  }
  if (do_cs || do_csa) {             // 16 row groups per column quad, summed in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(s_tn);   // [2][16 row groups][128 columns]
    *reinterpret_cast<f32x4*>(red + srow * 128 + 4 * c4) = cs;
    *reinterpret_cast<f32x4*>(red + 2048 + srow * 128 + 4 * c4) = csa;
    __syncthreads();
    if (tid < 256) {
      const int which = tid >> 7, c = tid & 127;
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += red[which * 2048 + g * 128 + c];
      if (which == 0 && do_cs) p.colsum_slab[(int64_t)z * p.No + j0 + c] = sum;
      if (which == 1 && do_csa) p.colsum_a_slab[(int64_t)z * p.Mo + i0 + c] = sum;
    }
  }
  float* slab = p.slab + (int64_t)z * p.Mo * p.No;
  const int col = j0 + 32 * wn + r;
#pragma unroll
  for (int bi = 0; bi < 2; ++bi)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = i0 + 32 * (2 * wm + bi) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      slab[(int64_t)row * p.No + col] = acc[bi][reg];
    }
}

// 128 x 128 tiles when both output dimensions allow it (B4R_TN_WIDE=0: the 64 x 64 kernel everywhere)
inline bool tn_wide_tiles(int Mo, int No) {
  static const int on = getenv("B4R_TN_WIDE") ? atoi(getenv("B4R_TN_WIDE")) : 1;
  return on && Mo >= 128 && No >= 128 && Mo % 128 == 0 && No % 128 == 0;
}

constexpr int TN_KS = 64;   // measured on the ML-1M shapes: 32 -> 1.185, 64 -> 1.173, 128 -> 1.197 ms/step

inline int tn_single_tile_cap() {
  static const int c = getenv("B4R_TN_CAP1") ? atoi(getenv("B4R_TN_CAP1")) : 512;
  return c;
}
int rx_tn_split(int R, int Mo, int No) {
  static const int wg_target = getenv("B4R_TN_TARGET") ? atoi(getenv("B4R_TN_TARGET")) : 512;
  static const int wg_target_wide = getenv("B4R_TN_TARGET_WIDE") ? atoi(getenv("B4R_TN_TARGET_WIDE")) : 256;   // measured at ML-20M: 128 -> 10.83, 256 -> 9.55, 384 -> 9.54, 512 -> 9.56 ms per step (64 x 64 tiles: 9.85)
  const bool wide = tn_wide_tiles(Mo, No);
  const int tw = wide ? 128 : 64;
  const int tiles = b4r_cdiv(Mo, tw) * b4r_cdiv(No, tw);
  int S = b4r_cdiv(wide ? wg_target_wide : wg_target, tiles);
  const int max_s = b4r_cdiv(R, 2 * TN_KS);  // at least two chunks per workgroup
  if (S > max_s) S = max_s;
  const int cap = tiles == 1 ? tn_single_tile_cap() : 256;   // one 64 x 64 output tile: the slices are the only parallelism
  if (S > cap) S = cap;
  if (S < 1) S = 1;
  return S;
}

inline bool vec_ok(const void* ptr, int ld) { return ptr != nullptr && b4r_aligned16(ptr) && (ld % 4 == 0); }
inline int up4i(int x) { return (x + 3) & ~3; }

}  // namespace

// shape contract of the branch-free kernels; everything else runs on the exact-fp32 LDS-tiled path
bool b4r_gemm_rx_supported(const b4r_gemm_desc* d) {
  const bool k_small = (d->K == 64 || d->K == 32);
  const bool k_loop = (d->K > 64 && d->K % 64 == 0);   // rx_gemm_kloop_kernel (passes of 64 or 128 columns)
  if (!(k_small || k_loop) || d->M % 32 != 0 || d->N < 4) return false;
  if (!vec_ok(d->A, d->lda) || !vec_ok(d->C, d->ldc)) return false;
  if (d->b_is_nk && !vec_ok(d->B, d->ldb)) return false;
  const int ns = up4i(d->N);
  // writing the pad columns [N, ns) is only allowed when the caller says they are scratch (c_pad_scratch) or there are none
  if (ns != d->N && (!d->c_pad_scratch || ns > d->ldc)) return false;
  const int epi = d->epilogue;
  if (epi_has_r(epi) && (!vec_ok(d->R, d->ldr) || ns > d->ldr)) return false;
  if (epi == B4R_EPI_BIAS_GELU && (!vec_ok(d->C2, d->ldc2) || ns > d->ldc2)) return false;
  if (epi == B4R_EPI_BIAS_DROP_RES_LN || epi == B4R_EPI_BIAS_GELU_LN) {
    if (d->N != 64 || d->K < 64 || d->K % WIDE_KC != 0 || d->b_is_nk || d->a_dropout) return false;
    if (!vec_ok(d->C2, d->ldc2) || d->ldc2 < 64 || !vec_ok(d->ln_gamma, 4) || !vec_ok(d->ln_beta, 4)) return false;
    if (epi == B4R_EPI_BIAS_GELU_LN && (!vec_ok(d->C3, d->ldc3) || d->ldc3 < 64)) return false;
  }
  if (epi == B4R_EPI_ADD_RES_LN_BWD) {
    if (d->N != 64 || d->K < 64 || d->K % WIDE_KC != 0 || !d->b_is_nk || d->a_dropout) return false;
    if (!vec_ok(d->C2, 4) || !vec_ok(d->ln_gamma, 4) || !d->ln_mean || !d->ln_rstd) return false;
    if (d->ln_ids) {
      if (!vec_ok(d->ln_table, 4) || !vec_ok(d->ln_pos, 4) || d->ln_L < 1 || d->ln_V < 1) return false;
    } else if (!vec_ok(d->ln_z, d->ln_ldz) || d->ln_ldz < 64) {
      return false;
    }
  }
  return true;
}

// called by b4r_gemm_f32 (argument checks already done there) in the bf16x3 mode when b4r_gemm_rx_supported
int b4r_gemm_rx_launch(const b4r_gemm_desc* d, hipStream_t stream) {
  RxP p;
  p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.C2 = d->C2; p.R = d->R;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldc2 = d->ldc2; p.ldr = d->ldr;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.n_store = up4i(d->N);
  p.qscale = d->qscale; p.qcols = d->qcols;
  p.ln_gamma = d->ln_gamma; p.ln_beta = d->ln_beta; p.ln_mean = d->ln_mean; p.ln_rstd = d->ln_rstd; p.ln_eps = d->ln_eps;
  p.ln_z = d->ln_z; p.ln_ldz = d->ln_ldz;
  p.ln_ids = d->ln_ids; p.ln_table = d->ln_table; p.ln_pos = d->ln_pos; p.ln_L = d->ln_L; p.ln_V = d->ln_V;
  p.C3 = d->C3; p.ldc3 = d->ldc3;
  p.a_idx = d->a_gather_idx; p.a_add_per = d->a_gather_add_per; p.a_per = d->a_gather_per > 0 ? d->a_gather_per : 1;
  p.a_copy = d->a_copy; p.a_copy_ld = d->a_copy_ld;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  p.k_chunks_per_split = d->K / 64 > 0 ? d->K / 64 : 1; p.slab_stride = 0;
  const bool a_drop = d->a_dropout && p.drop.rng != nullptr;
  const int mblocks = b4r_cdiv(d->M, 128);
  const int total_steps = b4r_cdiv(d->N, 32);
  // one resident round: 256 CUs x 4 workgroups (96 VGPRs -> 4 waves/SIMD).  A grid a little above that leaves a half-empty
  // second round (measured on the MLM-head projection: 960 workgroups 31.7 us, 1360 36.2 us, 3120 40.6 us), so round the
  // split count DOWN; B4R_RX_TARGET overrides the slot count for experiments
  static const int wg_slots = getenv("B4R_RX_TARGET") ? atoi(getenv("B4R_RX_TARGET")) : 1024;
  int splits = wg_slots / mblocks;
  if (splits > total_steps) splits = total_steps;
  if (splits < 1) splits = 1;
  p.steps_per_split = b4r_cdiv(total_steps, splits);
  p.n_splits = b4r_cdiv(total_steps, p.steps_per_split);
  p.n_items = mblocks * p.n_splits;
  dim3 grid(xcd_grid(p.n_items));
  // many column splits = a store-dominated product (the materialising vocabulary projection: 12 splits): there the
  // mapping concentrates each XCD's writes and costs time (34 -> 40 us measured), so it is kept to the few-split products
  if (!xcd_on() || p.n_splits > 4) p.n_items = -p.n_items;
  int rc = d->b_is_nk ? dispatch_rx<true>(p, d->epilogue, a_drop, grid, stream)
                      : dispatch_rx<false>(p, d->epilogue, a_drop, grid, stream);
  if (rc != B4R_OK) return rc;
  B4R_CHECK_LAUNCH("b4r_gemm_f32 (bf16x3)");
  return B4R_OK;
}

// plain product C = A.B with B as [K,N], N <= 128, reduced over `splits` ranges of K into slabs (the caller reduces them).
// k_pad_ok: the caller guarantees that columns [K, roundup(K,64)) of A are readable zeros and rows [K, roundup(K,64)) of B
// are readable finite values, so the reduction runs over whole chunks of 64.
bool b4r_gemm_rx_splitk_supported(const b4r_gemm_desc* d, int k_pad_ok) {
  const int Kp = (d->K + 63) & ~63;
  if (d->b_is_nk || d->epilogue != B4R_EPI_NONE || d->a_dropout) return false;
  if (Kp <= 64 || (Kp != d->K && (!k_pad_ok || d->lda < Kp))) return false;
  if (d->M % 32 != 0 || d->N % 4 != 0 || d->N > 128) return false;
  return vec_ok(d->A, d->lda) && d->B != nullptr;
}

int b4r_gemm_rx_splitk_launch(const b4r_gemm_desc* d, int splits, float* slabs, int* slabs_used, hipStream_t stream) {
  RxP p{};
  p.A = d->A; p.B = d->B; p.C = slabs; p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->N;
  p.M = d->M; p.N = d->N; p.K = (d->K + 63) & ~63;
  p.n_store = d->N;
  p.qscale = 1.f;
  p.drop = b4r_make_drop(nullptr, 0, 0.f, 0);
  const int chunks = p.K / 64;
  p.k_chunks_per_split = b4r_cdiv(chunks, splits < 1 ? 1 : splits);
  p.slab_stride = (int64_t)d->M * d->N;
  *slabs_used = b4r_cdiv(chunks, p.k_chunks_per_split);
  launch_kloop<false, B4R_EPI_NONE, false>(p, stream);
  B4R_CHECK_LAUNCH("gemm_splitk (bf16x3)");
  return B4R_OK;
}

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);

bool b4r_gemm_rx_tn_supported(const b4r_gemm_tn_desc* d) {
  if (!vec_ok(d->A, d->lda) || !vec_ok(d->B, d->ldb)) return false;
  return up4i(d->Mo) <= d->lda && up4i(d->No) <= d->ldb && d->Mo >= 4 && d->No >= 4;
}

int64_t b4r_gemm_rx_tn_scratch_floats(int R, int Mo, int No) {
  const int S = rx_tn_split(R, Mo, No);
  return (int64_t)S * Mo * No + (int64_t)S * No + (int64_t)S * Mo;
}

static RxTnP make_tn_params(const b4r_gemm_tn_desc* d, float* scratch, int S) {
  RxTnP p;
  p.A = d->A; p.B = d->B; p.lda = d->lda; p.ldb = d->ldb;
  p.R = d->R; p.Mo = d->Mo; p.No = d->No;
  const int tw = (tn_wide_tiles(d->Mo, d->No) && d->dgrad_out == nullptr) ? 128 : 64;
  p.tiles_i = b4r_cdiv(d->Mo, tw); p.tiles_j = b4r_cdiv(d->No, tw); p.S = S;
  p.chunk = b4r_cdiv(b4r_cdiv(d->R, S), TN_KS) * TN_KS;
  p.a_lim = up4i(d->Mo); p.b_lim = up4i(d->No);
  p.slab = scratch;
  p.colsum_slab = d->colsum ? scratch + (int64_t)S * d->Mo * d->No : nullptr;
  p.colsum_a_slab = d->colsum_a ? scratch + (int64_t)S * d->Mo * d->No + (int64_t)S * d->No : nullptr;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  p.W = d->dgrad_w; p.ldw = d->dgrad_ldw; p.dX = d->dgrad_out; p.lddx = d->dgrad_ldo; p.G = d->dgrad_gelu_pre; p.ldg = d->dgrad_ldg;
  return p;
}

int b4r_gemm_rx_tn_launch(const b4r_gemm_tn_desc* d, float* scratch, hipStream_t stream) {
  const int S = rx_tn_split(d->R, d->Mo, d->No);
  const RxTnP p = make_tn_params(d, scratch, S);
  const bool b_drop = d->b_dropout && p.drop.rng != nullptr;
  const int dgrad = d->dgrad_out == nullptr ? 0 : (d->dgrad_gelu_pre ? 2 : 1);   // b4r_gemm_tn_f32 has checked the shape contract
  const int64_t items = (int64_t)p.tiles_i * p.tiles_j * S;
  dim3 grid(xcd_grid(items));
  if (dgrad == 0 && tn_wide_tiles(d->Mo, d->No)) {
    constexpr size_t lds128 = (size_t)4 * TN128_KS * 256;   // 32 KB
    if (b_drop) hipLaunchKernelGGL((rx_gemm_tn128_kernel<true>), grid, dim3(512), lds128, stream, p);
    else hipLaunchKernelGGL((rx_gemm_tn128_kernel<false>), grid, dim3(512), lds128, stream, p);
    B4R_CHECK_LAUNCH("b4r_gemm_tn_f32 (bf16x3, 128 x 128 tiles)");
    return b4r_launch_slab_reduce_full(p.slab, S, d->Mo, d->No, d->out, d->ldo, d->accumulate, p.colsum_slab, d->colsum,
                                       p.colsum_a_slab, d->colsum_a, stream);
  }
  constexpr size_t lds = (size_t)4 * TN_KS * 128;
  static bool lds_raised = false;
  if (lds > 48 * 1024 && !lds_raised) {
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_kernel<TN_KS, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_kernel<TN_KS, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_kernel<TN_KS, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_kernel<TN_KS, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_kernel<TN_KS, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_kernel<TN_KS, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_raised = true;
  }
#define B4R_TN_LAUNCH(BD, DG) hipLaunchKernelGGL((rx_gemm_tn_kernel<TN_KS, BD, DG>), grid, dim3(256), lds, stream, p)
  if (dgrad == 2) { if (b_drop) B4R_TN_LAUNCH(true, 2); else B4R_TN_LAUNCH(false, 2); }
  else if (dgrad == 1) { if (b_drop) B4R_TN_LAUNCH(true, 1); else B4R_TN_LAUNCH(false, 1); }
  else if (b_drop) B4R_TN_LAUNCH(true, 0);
  else B4R_TN_LAUNCH(false, 0);
#undef B4R_TN_LAUNCH
  B4R_CHECK_LAUNCH("b4r_gemm_tn_f32 (bf16x3)");
  return b4r_launch_slab_reduce_full(p.slab, S, d->Mo, d->No, d->out, d->ldo, d->accumulate, p.colsum_slab, d->colsum,
                                     p.colsum_a_slab, d->colsum_a, stream);
}

// d0 with dropout on its B operand, d1 without, neither with an input-gradient tail: the two products of b4r_encoder_layer_bwd
bool b4r_gemm_rx_tn_pair_supported(const b4r_gemm_tn_desc* d0, const b4r_gemm_tn_desc* d1) {
  if (tn_wide_tiles(d0->Mo, d0->No) || tn_wide_tiles(d1->Mo, d1->No)) return false;   // those run rx_gemm_tn128_kernel, one launch each
  return b4r_gemm_rx_tn_supported(d0) && b4r_gemm_rx_tn_supported(d1) && !d0->dgrad_out && !d1->dgrad_out &&
         !(d1->b_dropout && d1->rng && d1->drop_rate > 0.f);
}
int b4r_gemm_rx_tn_pair_launch(const b4r_gemm_tn_desc* d0, float* scratch0, const b4r_gemm_tn_desc* d1, float* scratch1,
                               hipStream_t stream) {
  const int S0 = rx_tn_split(d0->R, d0->Mo, d0->No), S1 = rx_tn_split(d1->R, d1->Mo, d1->No);
  const RxTnP p0 = make_tn_params(d0, scratch0, S0), p1 = make_tn_params(d1, scratch1, S1);
  const bool drop0 = d0->b_dropout && p0.drop.rng != nullptr;
  const int n0 = (int)xcd_grid((int64_t)p0.tiles_i * p0.tiles_j * S0), n1 = (int)xcd_grid((int64_t)p1.tiles_i * p1.tiles_j * S1);
  constexpr size_t lds = (size_t)4 * TN_KS * 128;
  static bool lds_raised = false;
  if (lds > 48 * 1024 && !lds_raised) {
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_pair_kernel<TN_KS, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)rx_gemm_tn_pair_kernel<TN_KS, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_raised = true;
  }
  if (drop0) hipLaunchKernelGGL((rx_gemm_tn_pair_kernel<TN_KS, true, false>), dim3(n0 + n1), dim3(256), lds, stream, p0, p1, n0);
  else hipLaunchKernelGGL((rx_gemm_tn_pair_kernel<TN_KS, false, false>), dim3(n0 + n1), dim3(256), lds, stream, p0, p1, n0);
  B4R_CHECK_LAUNCH("b4r_gemm_tn_f32 pair (bf16x3)");
  int rc = b4r_launch_slab_reduce_full(p0.slab, S0, d0->Mo, d0->No, d0->out, d0->ldo, d0->accumulate, p0.colsum_slab, d0->colsum,
                                       p0.colsum_a_slab, d0->colsum_a, stream);
  if (rc) return rc;
  return b4r_launch_slab_reduce_full(p1.slab, S1, d1->Mo, d1->No, d1->out, d1->ldo, d1->accumulate, p1.colsum_slab, d1->colsum,
                                     p1.colsum_a_slab, d1->colsum_a, stream);
}
