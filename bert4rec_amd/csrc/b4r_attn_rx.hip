// Keras MultiHeadAttention core (head_dim 32), split-precision variant: every product runs on v_mfma_f32_16x16x32_bf16
// with both operands split into bf16 hi + lo (x = hi + lo; A.B ~= Alo.Bhi + Ahi.Blo + Ahi.Bhi, fp32 accumulate), i.e. about
// 2^-16 relative error per product at 5.3x the fp32-MFMA rate (b4r_gemm_rx.hip uses the same arithmetic).
//
// Same decomposition as the fp32 kernels of b4r_attn.hip (a wave owns 16 queries -- or 16 keys in the dK/dV kernel -- and
// sweeps the other dimension in 16-row tiles; the score tile is computed in the orientation whose accumulator ROWS are the
// index the following product sums over, so probabilities never move between lanes).  What changes:
//   * K and V (or Q and dO) live in LDS as bf16 hi / lo IMAGES of [rows][32] with 64-byte rows, written once per
//     workgroup.  One image serves both operand shapes: a row fragment (8 consecutive columns of one row, ds_read_b128)
//     for the products that sum over the head dimension, and a transposed fragment (one column of 4 consecutive rows per
//     lane, ds_read_b64_tr_b16) for the products that consume an accumulator tile.  The 16-byte chunk index of a row is
//     XORed with (-(row>>2))&3, which makes both reads bank-conflict free without padding (MI355X_MICROARCH.md, LDS).
//     The four images of a kernel are interleaved per 16-row tile ([tile][image][16 rows]) so that an image is a
//     compile-time offset and every LDS address of a loop iteration is one lane constant + one tile offset + immediate.
//   * a 16x16x32 product that consumes accumulator tiles takes TWO of them per instruction: k-slot (g, j) is row
//     16*t0 + 4g + j of tile t0 for j < 4 and row 16*t1 + 4g + (j-4) of tile t1 for j >= 4, on both operands.
//   * the forward kernel stores the dropout decisions it hashed as bits (4 per lane and key tile); the two backward
//     kernels read them back instead of hashing every (query, key) pair twice more.  Word layout: [b, head, query tile of
//     16, key-tile group of 8][forward lane]; nibble (t & 7) of the word holds keys 16t + 4g .. +3 of query (lane & 15).
#include "b4r_rx_tiles.h"

namespace {

// 1-D grid with XCD-aware logical ids (consecutive workgroup ids go round-robin over the 8 XCDs): the row blocks of one
// (batch, head) stage the same K / V (or Q / dO) rows, so they get consecutive logical ids = one XCD's L2.
struct WgCoord { int x, hd, b; bool live; };
__device__ __forceinline__ WgCoord wg_coord(int nx, int heads, int B) {
  const int n = nx * heads * B, per = (n + 7) >> 3;
  const int lid = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  WgCoord c;
  c.live = lid < n;
  const int l = c.live ? lid : 0;
  c.x = l % nx; c.hd = (l / nx) % heads; c.b = l / (nx * heads);
  return c;
}
inline dim3 wg_grid(int nx, int heads, int B) { return dim3((unsigned)((((int64_t)nx * heads * B + 7) >> 3) << 3)); }

struct AttnRxP {
  const float* qkv; const int64_t* mask; const float* ctx; const float* lse_in; const float* dctx;
  float* ctx_out; float* lse_out; float* dqkv;
  uint32_t* bits_out; const uint32_t* bits_in;
  uint32_t* bits32;   // forward: the same decisions once more as one word per (query, 32-key tile), what b4r_attn32.hip's backward reads (or NULL)
  int B, L, heads, H;
  int KT, KTE;   // 16-row tiles covering L, and KT rounded up to even (the images hold KTE tiles, zero beyond L)
  float qscale;
  DropArgs drop;
};

// D[r] = sum_c dO[r][c] * O[r][c] over the 32 columns of this head, for rows [0,nrows), nrows <= 256; rows beyond valid -> 0
__device__ __forceinline__ void rowdot_head(float* sD, const float* dO, const float* O, int64_t row0, int ld, int nrows, int valid) {
  const int part = threadIdx.x & 3;
  f32x4 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = min(16 * WAVES * it + (int)(threadIdx.x >> 2), valid - 1);
    const float* a = dO + (row0 + r) * ld + part * 8;
    const float* b = O + (row0 + r) * ld + part * 8;
    a0[it] = *reinterpret_cast<const f32x4*>(a); a1[it] = *reinterpret_cast<const f32x4*>(a + 4);
    b0[it] = *reinterpret_cast<const f32x4*>(b); b1[it] = *reinterpret_cast<const f32x4*>(b + 4);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = 16 * WAVES * it + (int)(threadIdx.x >> 2);
    float s = (a0[it][0] * b0[it][0] + a0[it][1] * b0[it][1]) + (a0[it][2] * b0[it][2] + a0[it][3] * b0[it][3]) +
              (a1[it][0] * b1[it][0] + a1[it][1] * b1[it][1]) + (a1[it][2] * b1[it][2] + a1[it][3] * b1[it][3]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (part == 0 && r < nrows) sD[r] = r < valid ? s : 0.f;
  }
}

// -----------------------------------------------------------------------------------------------------------
// forward: workgroup = 128 queries of one (batch, head); wave = 16 queries x all keys
// LDS: [K hi | K lo | V hi | V lo | sAdd]
// -----------------------------------------------------------------------------------------------------------
template <int KT>
__global__ __launch_bounds__(64 * WAVES) void attn_rx_fwd_kernel(AttnRxP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_rx[];
  constexpr int KTE = (KT + 1) & ~1, LPE = KTE * 16;
  char* img = smem_rx;                                    // images 0/1 = K hi/lo, 2/3 = V hi/lo
  float* sAdd = reinterpret_cast<float*>(img + KTE * TILE_BYTES);

  const WgCoord wg = wg_coord((p.L + ROWS_WG - 1) / ROWS_WG, p.heads, p.B);
  if (!wg.live) return;   // block-uniform, before any barrier
  const int b = wg.b, hd = wg.hd, q0 = wg.x * ROWS_WG;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const float amax = b4r_seq_amax(p.mask + row0, L);

  const int q = q0 + 16 * wave + i;
  const f32x8 qx = load8(p.qkv + (row0 + min(q, L - 1)) * ld3 + hd * 32 + 8 * g);   // in flight while K / V are staged
  StagedRows st;
  stage_fetch(st, p.qkv + H + hd * 32, ld3, p.qkv + 2 * H + hd * 32, ld3, row0, L);
  const int kk = min((int)threadIdx.x, L - 1);   // LPE <= 256 < workgroup size
  const float madd = (1.0f - (float)p.mask[row0 + kk]) * -1e9f;
  stage_write(st, img, LPE, L);
  if (threadIdx.x < LPE) sAdd[threadIdx.x] = (int)threadIdx.x < L ? madd : -INFINITY;
  bf16x8 qh, ql;
  split8(qx, qh, ql);
  __syncthreads();
  if (q0 + 16 * wave >= L) return;  // wave-uniform; no barrier below, and EXEC stays full for the transposed reads

  const FragAddr fa = frag_addr(lane);
  f32x4 acc[KTE];
#pragma unroll
  for (int t = 0; t < KTE; ++t) {
    acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* tile = img + fa.row + TILE_BYTES * t;
    if (t < KT) acc[t] = mfma3(row_frag<0>(tile), row_frag<1>(tile), qh, ql, acc[t]);   // S^T = K.Q^T
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
  // (score - max) first: with Keras' -1e9 mask a fully masked row has scores and max of magnitude 1e9, and folding log2(e)
  // into separately rounded terms would lose the exact cancellation (same in the backward kernels)
  f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const f32x4 d = (acc[t] - m) * 1.4426950408889634f;   // packed subtract / multiply; e^x = 2^(x log2 e) as __expf does
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_amdgcn_exp2f(d[r]);
    sum4 += acc[t];
  }
  sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const int64_t bh = (int64_t)b * p.heads + hd;
  if (g == 0 && q < L && p.lse_out) p.lse_out[bh * L + q] = (m - amax) + __logf(sum);

  const DropCtx dctx = b4r_drop_ctx(p.drop);
  if (dctx.on) {
    const uint64_t dbase = ((uint64_t)bh * L + (uint64_t)(q < L ? q : 0)) * (uint64_t)B4R_ATTN_PITCH;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const B4rKeep4 k4 = b4r_keep4p(dctx, dbase + (uint64_t)(16 * t + 4 * g));   // one hash for the lane's 4 keys
      const f32x4 ps = acc[t] * (inv * dctx.scale);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[t][s] = k4.k[s] ? ps[s] : 0.f;
      w[t >> 3] |= k4.bits() << (4 * (t & 7));
    }
    uint32_t* wo = p.bits_out + ((bh * p.KT + (q0 >> 4) + wave) * 2) * 64 + lane;
    wo[0] = w[0];
    wo[64] = w[1];
    if (p.bits32) {   // [b][head][key tile T][query tile][16 register pairs][2]: bit k of the word = key 32 T + k
      const int NT = (L + 31) >> 5, qt = q >> 5, qr = q & 31;
      const int slot = (qr & 24) | ((qr & 3) << 1) | ((qr >> 2) & 1);   // query 16s + 8a + 4h' + b -> 16s + 8a + 2b + h'
      for (int T = 0; T < NT; ++T) {
        const uint32_t by = (w[T >> 2] >> (8 * (T & 3))) & 0xFFu;        // the nibbles of the 16-key tiles 2T, 2T + 1
        uint32_t part = ((by & 15u) | ((by >> 4) << 16)) << (4 * g);     // rows 4g .. 4g+3 of each
        part |= (uint32_t)__shfl_xor((int)part, 16, 64);
        part |= (uint32_t)__shfl_xor((int)part, 32, 64);
        if (g == 0 && q < L) p.bits32[((bh * NT + T) * NT + qt) * 32 + slot] = part;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < KT; ++t) acc[t] = acc[t] * inv;
  }

  f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int tp = 0; tp < KTE / 2; ++tp) {   // O^T[dd][query] += V^T[dd][keys of two tiles] . P^T[keys][query]
    bf16x8 ph, pl;
    split8(cat(acc[2 * tp], acc[2 * tp + 1]), ph, pl);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const char* tile = img + fa.tr[db] + TILE_BYTES * 2 * tp;
      o[db] = mfma3(tr_frag<2>(tile), tr_frag<3>(tile), ph, pl, o[db]);
    }
  }
  if (q < L) {
    float* dst = p.ctx_out + (row0 + q) * H + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(dst) = o[0];
    *reinterpret_cast<f32x4*>(dst + 16) = o[1];
  }
}

// -----------------------------------------------------------------------------------------------------------
// backward, dQ: same decomposition as the forward; probabilities recomputed from the saved log-sum-exp
// LDS: [K hi | K lo | V hi | V lo | sAdd | sD]
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void attn_rx_dq_kernel(AttnRxP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_rx[];
  const int KTE = p.KTE, LPE = KTE * 16;
  char* img = smem_rx;                                    // images 0/1 = K hi/lo, 2/3 = V hi/lo
  float* sAdd = reinterpret_cast<float*>(img + KTE * TILE_BYTES);
  float* sD = sAdd + LPE;

  const WgCoord wg = wg_coord((p.L + ROWS_WG - 1) / ROWS_WG, p.heads, p.B);
  if (!wg.live) return;   // block-uniform, before any barrier
  const int b = wg.b, hd = wg.hd, q0 = wg.x * ROWS_WG;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const float amax = b4r_seq_amax(p.mask + row0, L);

  const int q = q0 + 16 * wave + i;
  const bool qlive = q < L;
  const int qc = min(q, L - 1);
  const f32x8 qx = load8(p.qkv + (row0 + qc) * ld3 + hd * 32 + 8 * g);
  const f32x8 dox = load8(p.dctx + (row0 + qc) * H + hd * 32 + 8 * g);
  StagedRows st;
  stage_fetch(st, p.qkv + H + hd * 32, ld3, p.qkv + 2 * H + hd * 32, ld3, row0, L);
  const int kk = min((int)threadIdx.x, L - 1);   // LPE <= 256 < workgroup size
  const float madd = (1.0f - (float)p.mask[row0 + kk]) * -1e9f;
  rowdot_head(sD, p.dctx + hd * 32, p.ctx + hd * 32, row0 + q0, H, ROWS_WG, L - q0);
  stage_write(st, img, LPE, L);
  if (threadIdx.x < LPE) sAdd[threadIdx.x] = (int)threadIdx.x < L ? madd : -INFINITY;
  bf16x8 qh, ql, doh, dol;
  split8(qx, qh, ql);
  split8(dox, doh, dol);
  // per-lane scalars of the sweep are requested before the barrier too (they used to cost a round trip after it)
  const int64_t bh = (int64_t)b * p.heads + hd;
  const float lse = p.lse_in[bh * L + qc];
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  uint32_t w[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
  if (dctx.on) {
    const int qt = min((q0 >> 4) + wave, p.KT - 1);             // waves beyond L exit below; keep their address in range
    const uint32_t* wi = p.bits_in + ((bh * p.KT + qt) * 2) * 64 + lane;
    w[0] = wi[0];
    w[1] = wi[64];
  }
  const float dscale = dctx.on ? dctx.scale : 1.0f;
  __syncthreads();
  if (q0 + 16 * wave >= L) return;

  const FragAddr fa = frag_addr(lane);
  const float Dq = sD[16 * wave + i];

  f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int tp = 0; tp < KTE / 2; ++tp) {
    f32x4 ds[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * tp + u;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const char* tile = img + fa.row + TILE_BYTES * t;
      const f32x4 sc = mfma3(row_frag<0>(tile), row_frag<1>(tile), qh, ql, z);     // S^T = K.Q^T
      const f32x4 da = mfma3(row_frag<2>(tile), row_frag<3>(tile), doh, dol, z);   // dA^T = V.dO^T
      const f32x4 ad = *reinterpret_cast<const f32x4*>(&sAdd[16 * t + 4 * g]);
      const uint32_t nib = w[(t >> 3) & 1] >> (4 * (t & 7));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float pr = __expf(((sc[s] + ad[s]) - amax) - lse);
        const float dA = ((nib >> s) & 1u) ? da[s] * dscale : 0.f;
        ds[u][s] = pr * (dA - Dq);
      }
    }
    bf16x8 dsh, dsl;
    split8(cat(ds[0], ds[1]), dsh, dsl);
#pragma unroll
    for (int db = 0; db < 2; ++db) {   // dQ^T[dk][query] += K^T[dk][keys] . dS^T[keys][query]
      const char* tile = img + fa.tr[db] + TILE_BYTES * 2 * tp;
      dq[db] = mfma3(tr_frag<0>(tile), tr_frag<1>(tile), dsh, dsl, dq[db]);
    }
  }
  if (qlive) {
    float* dst = p.dqkv + (row0 + q) * ld3 + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(dst) = dq[0] * p.qscale;
    *reinterpret_cast<f32x4*>(dst + 16) = dq[1] * p.qscale;
  }
}

// -----------------------------------------------------------------------------------------------------------
// backward, dK / dV: workgroup = 128 keys of one (batch, head); wave = 16 keys x all queries
// LDS: [Q hi | Q lo | dO hi | dO lo | sLse | sD]
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES) void attn_rx_dkv_kernel(AttnRxP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_rx[];
  const int KTE = p.KTE, LPE = KTE * 16;
  char* img = smem_rx;                                    // images 0/1 = Q hi/lo, 2/3 = dO hi/lo
  float* sLse = reinterpret_cast<float*>(img + KTE * TILE_BYTES);
  float* sD = sLse + LPE;

  const WgCoord wg = wg_coord((p.L + ROWS_WG - 1) / ROWS_WG, p.heads, p.B);
  if (!wg.live) return;   // block-uniform, before any barrier
  const int b = wg.b, hd = wg.hd;
  const int L = p.L, H = p.H, ld3 = 3 * H;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const int64_t bh = (int64_t)b * p.heads + hd;
  const float amax = b4r_seq_amax(p.mask + row0, L);

  const int k0 = (wg.x * WAVES + wave) * 16;
  const int key = k0 + i;
  const bool klive = key < L;
  const int kc = min(key, L - 1);
  const f32x8 kx = load8(p.qkv + (row0 + kc) * ld3 + H + hd * 32 + 8 * g);
  const f32x8 vx = load8(p.qkv + (row0 + kc) * ld3 + 2 * H + hd * 32 + 8 * g);
  StagedRows st;
  stage_fetch(st, p.qkv + hd * 32, ld3, p.dctx + hd * 32, H, row0, L);
  const float lse_k = p.lse_in[bh * L + min((int)threadIdx.x, L - 1)];   // LPE <= 256 < workgroup size
  rowdot_head(sD, p.dctx + hd * 32, p.ctx + hd * 32, row0, H, LPE, L);
  stage_write(st, img, LPE, L);
  if (threadIdx.x < LPE) sLse[threadIdx.x] = (int)threadIdx.x < L ? lse_k : INFINITY;   // +inf => probability 0 for pad queries
  bf16x8 kh, kl, vh, vl;
  split8(kx, kh, kl);
  split8(vx, vh, vl);
  const float kmask = (float)p.mask[row0 + kc];                // requested before the barrier
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  __syncthreads();
  if (k0 >= L) return;  // wave-uniform; no barrier below

  const FragAddr fa = frag_addr(lane);
  const float add = klive ? (1.0f - kmask) * -1e9f : -INFINITY;
  const float dscale = dctx.on ? dctx.scale : 1.0f;
  const int tk = k0 >> 4;
  // the forward lane that hashed (query 16t + 4g + r, this key) is lane (4g + r) + 16 * (i >> 2): 4 consecutive words
  const uint32_t* wbase = p.bits_in + ((bh * p.KT) * 2 + (tk >> 3)) * 64 + 4 * g + 16 * (i >> 2);
  const int wshift = 4 * (tk & 7) + (i & 3);

  f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // software pipeline: the two products that only need LDS operands (S, dA) and the dropout words of pair tp+1 are
  // issued before the exp / select / split work of pair tp, so the matrix pipe and the LDS reads overlap that VALU work
  // inside one wave (3 waves per SIMD cannot hide them otherwise).  The look-ahead of the last pair recomputes itself.
  f32x4 sc_n[2], da_n[2];
  u32x4 wq_n[2] = {{~0u, ~0u, ~0u, ~0u}, {~0u, ~0u, ~0u, ~0u}};
  auto lookahead = [&](int tpn) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * tpn + u;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (dctx.on)   // query tiles beyond the forward's (odd KT) hold only pad queries: any word will do
        wq_n[u] = *reinterpret_cast<const u32x4*>(wbase + (int64_t)min(t, p.KT - 1) * 128);
      const char* tile = img + fa.row + TILE_BYTES * t;
      sc_n[u] = mfma3(row_frag<0>(tile), row_frag<1>(tile), kh, kl, z);   // S = Q.K^T
      da_n[u] = mfma3(row_frag<2>(tile), row_frag<3>(tile), vh, vl, z);   // dA = dO.V^T
    }
  };
  lookahead(0);
  const int NP = KTE / 2;
  for (int tp = 0; tp < NP; ++tp) {
    f32x4 pd[2], ds[2];
    const f32x4 sc[2] = {sc_n[0], sc_n[1]}, da[2] = {da_n[0], da_n[1]};
    const u32x4 wq[2] = {wq_n[0], wq_n[1]};
    lookahead(min(tp + 1, NP - 1));
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = 2 * tp + u;
      const f32x4 ls = *reinterpret_cast<const f32x4*>(&sLse[16 * t + 4 * g]);
      const f32x4 dd = *reinterpret_cast<const f32x4*>(&sD[16 * t + 4 * g]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __expf(((sc[u][r] + add) - amax) - ls[r]);
        const bool keep = (wq[u][r] >> wshift) & 1u;
        pd[u][r] = keep ? pr * dscale : 0.f;
        const float dA = keep ? da[u][r] * dscale : 0.f;
        ds[u][r] = pr * (dA - dd[r]);
      }
    }
    bf16x8 pdh, pdl, dsh, dsl;
    split8(cat(pd[0], pd[1]), pdh, pdl);
    split8(cat(ds[0], ds[1]), dsh, dsl);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      // dV^T[dd][key] += dO^T[dd][queries] . Pd[queries][key] ;  dK^T[dk][key] += Q^T[dk][queries] . dS[queries][key]
      const char* tile = img + fa.tr[db] + TILE_BYTES * 2 * tp;
      dv[db] = mfma3(tr_frag<2>(tile), tr_frag<3>(tile), pdh, pdl, dv[db]);
      dk[db] = mfma3(tr_frag<0>(tile), tr_frag<1>(tile), dsh, dsl, dk[db]);
    }
  }
  if (klive) {
    float* ok = p.dqkv + (row0 + key) * ld3 + H + hd * 32 + 4 * g;
    float* ov = p.dqkv + (row0 + key) * ld3 + 2 * H + hd * 32 + 4 * g;
    *reinterpret_cast<f32x4*>(ok) = dk[0];
    *reinterpret_cast<f32x4*>(ok + 16) = dk[1];
    *reinterpret_cast<f32x4*>(ov) = dv[0];
    *reinterpret_cast<f32x4*>(ov + 16) = dv[1];
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes) { return b4r_raise_lds((const void*)kernel, bytes, "attention"); }

}  // namespace

// dropout decisions of one layer: [B, heads, ceil(L/16) query tiles, 2 words, 64 lanes] uint32
int64_t b4r_attn_rx_keep_words(int B, int L, int heads) { return (int64_t)B * heads * b4r_cdiv(L, 16) * 128; }

// called by b4r_attn_fwd / b4r_attn_bwd (argument checks already done there) in the bf16x3 mode
// b4r_attn32.hip: the core on 32-token tiles (one workgroup per sequence and head, one launch each way), preferred from L = 65 to 224
bool b4r_attn32_core_preferred(int L);
bool b4r_attn32_core_fwd_wanted();
int64_t b4r_attn_rx_keep_words(int B, int L, int heads);
int b4r_attn32_core_fwd_launch(const float* qkv, const int64_t* mask, int B, int L, int heads, float* ctx, float* lse,
                               const DropArgs& drop, uint32_t* keep_bits, hipStream_t stream);
int b4r_attn32_core_bwd_launch(const float* qkv, const int64_t* mask, const float* ctx, const float* lse, const float* dctx, int B,
                               int L, int heads, float qscale, float* dqkv, const DropArgs& drop, const uint32_t* keep_bits,
                               hipStream_t stream);

int b4r_attn_rx_fwd_launch(const float* qkv, const int64_t* mask, int B, int L, int heads, float* ctx, float* lse,
                           const DropArgs& drop, uint32_t* keep_bits, hipStream_t stream) {
  // The forward stays on the 16-token-tile kernel below (ML-20M shape: 110 us against 133 for the 32-token-tile core, whose 112
  // score registers per lane leave one workgroup per CU) and writes the dropout decisions in the backward core's layout as well;
  // B4R_ATTN32_CORE_FWD=1 / b4r_attn32_set_core_fwd selects the 32-token-tile forward.
  if (b4r_attn32_core_preferred(L) && b4r_attn32_core_fwd_wanted())
    return b4r_attn32_core_fwd_launch(qkv, mask, B, L, heads, ctx, lse, drop, keep_bits, stream);
  AttnRxP p{};
  p.qkv = qkv; p.mask = mask; p.ctx_out = ctx; p.lse_out = lse; p.bits_out = keep_bits;
  if (keep_bits != nullptr && b4r_attn32_core_preferred(L)) p.bits32 = keep_bits + b4r_attn_rx_keep_words(B, L, heads);
  p.B = B; p.L = L; p.heads = heads; p.H = heads * 32;
  p.KT = b4r_cdiv(L, 16);
  p.drop = drop;
  // the register-resident score row is a compile-time number of tiles; unused tiles cost one masked product each
  const int KTt = p.KT <= 4 ? 4 : p.KT <= 8 ? 8 : p.KT <= 13 ? 13 : 16;
  p.KT = b4r_cdiv(L, 16);
  p.KTE = (KTt + 1) & ~1;
  const size_t sh = (size_t)4 * p.KTE * 16 * 64 + (size_t)p.KTE * 16 * sizeof(float);
  const dim3 grid = wg_grid(b4r_cdiv(L, ROWS_WG), heads, B);
  int rc;
#define FWD_CASE(KT_)                                                                                      \
  case KT_:                                                                                                \
    rc = set_lds(attn_rx_fwd_kernel<KT_>, sh);                                                             \
    if (rc) return rc;                                                                                     \
    hipLaunchKernelGGL((attn_rx_fwd_kernel<KT_>), grid, dim3(64 * WAVES), sh, stream, p);                  \
    break;
  switch (KTt) {
    FWD_CASE(4) FWD_CASE(8) FWD_CASE(13) FWD_CASE(16)
    default: b4r_set_error("b4r_attn_fwd: internal"); return B4R_E_SHAPE;
  }
#undef FWD_CASE
  B4R_CHECK_LAUNCH("b4r_attn_fwd (bf16x3)");
  return B4R_OK;
}

int b4r_attn_rx_bwd_launch(const float* qkv, const int64_t* mask, const float* ctx, const float* lse, const float* dctx,
                           int B, int L, int heads, float qscale, float* dqkv, const DropArgs& drop,
                           const uint32_t* keep_bits, hipStream_t stream, hipStream_t stream_dkv) {
  // stream_dkv: the dK/dV kernel may run on another stream (it writes other columns of dqkv than dQ does)
  if (b4r_attn32_core_preferred(L))   // one launch forms dq, dk and dv (on `stream`: callers that split the streams order them)
    return b4r_attn32_core_bwd_launch(qkv, mask, ctx, lse, dctx, B, L, heads, qscale, dqkv, drop, keep_bits, stream);
  AttnRxP p{};
  p.qkv = qkv; p.mask = mask; p.ctx = ctx; p.lse_in = lse; p.dctx = dctx; p.dqkv = dqkv; p.bits_in = keep_bits;
  p.B = B; p.L = L; p.heads = heads; p.H = heads * 32; p.qscale = qscale;
  p.KT = b4r_cdiv(L, 16);
  p.KTE = (p.KT + 1) & ~1;
  p.drop = drop;
  const dim3 grid = wg_grid(b4r_cdiv(L, ROWS_WG), heads, B);
  const size_t planes = (size_t)4 * p.KTE * 16 * 64;
  const size_t sh_dq = planes + ((size_t)p.KTE * 16 + ROWS_WG) * sizeof(float);
  int rc = set_lds(attn_rx_dq_kernel, sh_dq);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_rx_dq_kernel, grid, dim3(64 * WAVES), sh_dq, stream, p);
  B4R_CHECK_LAUNCH("b4r_attn_bwd dq (bf16x3)");
  const size_t sh_kv = planes + (size_t)2 * p.KTE * 16 * sizeof(float);
  rc = set_lds(attn_rx_dkv_kernel, sh_kv);
  if (rc) return rc;
  hipLaunchKernelGGL(attn_rx_dkv_kernel, grid, dim3(64 * WAVES), sh_kv, stream_dkv ? stream_dkv : stream, p);
  B4R_CHECK_LAUNCH("b4r_attn_bwd dkv (bf16x3)");
  return B4R_OK;
}
