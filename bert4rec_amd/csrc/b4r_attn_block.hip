// Attention half of a Keras TransformerEncoderBlock (post-LN) as ONE launch, one workgroup per sequence (hidden size 64,
// two heads of 32, L <= 256, split-precision bf16x3 arithmetic):
//
//   x [L,64] -> q,k,v = x.Wqkv + b (q scaled) -> per head softmax(q k^T + key mask) -> dropout -> . v -> ctx
//            -> ctx.Wo + bo -> dropout -> + x -> LayerNorm -> x1          (also z1, the LayerNorm statistics, ctx, lse,
//                                                                           the dropout decisions and -- optionally -- qkv)
//
// Reference: tfm TransformerEncoderBlock / Keras MultiHeadAttention as constructed at
// bert4rec/models/components/networks/bert4rec_encoder.py:136-147 and called at :220-222 (key-padding
// mask -1e9, query scaled by 1/sqrt(d) after its bias, attention dropout on the probabilities, output dropout, residual,
// self_attention_layer_norm).
//
// Round 1 ran this as three launches (QKV product 19 us, attention 29 us, output product + LayerNorm 14 us at ML-1M) that
// moved qkv (39 MB) and ctx through HBM twice; here a sequence never leaves its CU between x and x1.  Wave w owns token tile
// w (16 tokens) in every phase:
//   1. q,k,v^T = Wqkv^T.x^T for its tokens (transposed orientation of b4r_ffn_rx.hip: tokens on the lane index, so the
//      result tiles are B operands / 8-byte image pieces without any lane movement); q stays in registers.
//   2. after a barrier (Wqkv's LDS image is dead) k, v go into the attention images of b4r_rx_tiles.h, which overlay it.
//   3. per head the body of attn_rx_fwd_kernel (b4r_attn_rx.hip): S^T = K.Q^T in registers, softmax, dropout bits, O^T = V^T.P^T.
//   4. y^T = Wo^T.ctx^T straight from the two heads' accumulators, bias, dropout, residual, LayerNorm (two 4-lane shuffles).
// LDS: Wo image 16 KB + max(Wqkv image 48 KB, K / V images of both heads: 8 KB per 16 tokens) + the key mask.
#include "b4r_block_tiles.h"

namespace {

struct AbP {
  const float* x; const int64_t* mask;
  const float* Wqkv; const float* bqkv; const float* Wo; const float* bo; const float* g1; const float* be1;
  float* qkv; float* ctx; float* lse; uint32_t* bits;
  uint32_t* bits32;   // the same decisions as one word per (query, 32-key tile), the layout b4r_attn32.hip's backward reads (or NULL)
  float* z1; float* x1; float* mean1; float* rstd1;
  int B, L, KT;
  float qscale, eps;
  DropArgs drop_p, drop_o;
  // first layer (ids != NULL): the block input is formed here, x = dropout(LayerNorm(table[id] + pos[position])) (the embedding
  // stage, bert4rec_encoder.py:198-214), and written to x_out with its statistics
  const int64_t* ids; const float* table; const float* pos; const float* g0; const float* be0;
  float* x_out; float* mean0; float* rstd0;
  int V; float eps0;
  DropArgs drop_e;
};

__device__ __forceinline__ f32x4 lo4(const f32x8 v) { return (f32x4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ f32x4 hi4(const f32x8 v) { return (f32x4){v[4], v[5], v[6], v[7]}; }

// KTT: compile-time bound of the 16-key tiles (the score row lives in registers); block = 64 * KT threads, KT = ceil(L / 16)
// AB_PROF (tools/build_variant.sh ... -DAB_PROF): wave 0 of workgroup 0 stamps the shader clock at phase boundaries
#ifdef AB_PROF
__device__ long long g_ab_prof[64];
__device__ long long g_af_prof[16];
#define AF_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_af_prof[k] = clock64(); } while (0)
#define AB_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ab_prof[k] = clock64(); } while (0)
#else
#define AB_MARK(k) do { } while (0)
#define AF_MARK(k) do { } while (0)
#endif
template <int KTT>
__global__ __launch_bounds__(1024) void attn_block_fwd_kernel(AbP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_ab[];
  constexpr int KTE = (KTT + 1) & ~1;
  constexpr int KV_BYTES = 2 * KTE * TILE_BYTES;                       // both heads: [head][tile][K hi, K lo, V hi, V lo]
  constexpr int BIG = KV_BYTES > 48 * 1024 ? KV_BYTES : 48 * 1024;
  char* woimg = smem_ab;                      // [4 rt][2 cb] x (hi, lo): 16 KB
  char* big = smem_ab + 16 * 1024;            // Wqkv image [4 rt][6 cb] x (hi, lo) = 48 KB, then the K / V images
  float* sAdd = reinterpret_cast<float*>(big + BIG);   // [KTE * 16]
  float* sbq = sAdd + KTE * 16;               // bqkv [192]

  const int nthreads = blockDim.x;
  const int b = blockIdx.x, L = p.L;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4, qq = i >> 2, pp = i & 3;

  // ---- phase 0: weights and the key mask ------------------------------------------------------------------------------
  AF_MARK(0);
  const int tok = 16 * wave + i, tokc = min(tok, L - 1);
  // b4r_seq_amax, its barrier folded into the one behind the weight staging
  // the workgroup has at least 4 threads per token: one turn covers the key mask; it and the bias are requested here, ahead of the
  // weight staging, and go to LDS behind it (three load latencies less in front of the first barrier)
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  const bool bias_one_turn = blockDim.x >= 3 * HID;
  const float bq_val = (bias_one_turn && threadIdx.x < 3 * HID) ? p.bqkv[threadIdx.x] : 0.f;
  f32x8 xv[2];
  const float* xsrc = p.x;   // the residual of the epilogue is read from here
  float emb_mean = 0.f, emb_rstd = 0.f;
  if (p.ids != nullptr) {   // block-uniform: the embedding stage for this wave's tokens (position = token index in the sequence)
    int64_t id = p.ids[row0 + tokc];
    if (id < 0 || id >= p.V) id = 0;   // out-of-range ids read the PAD row, as b4r_embed_ln_fwd does
    float s0 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      xv[ks] = load8(p.table + id * HID + 32 * ks + 8 * g) + load8(p.pos + (int64_t)tokc * HID + 32 * ks + 8 * g);
#pragma unroll
      for (int e = 0; e < 8; ++e) s0 += xv[ks][e];
    }
    const float mean = quad_sum(s0) * (1.0f / HID);
    float q0 = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = xv[ks][e] - mean; q0 += d * d; }
    const float rstd = rsqrtf(quad_sum(q0) * (1.0f / HID) + p.eps0);
    const DropCtx dce = b4r_drop_ctx(p.drop_e);
    const bool live0 = tok < L;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const f32x8 gm = load8(p.g0 + 32 * ks + 8 * g), be = load8(p.be0 + 32 * ks + 8 * g);
      f32x8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float inv = rstd * gm[e];
        y[e] = xv[ks][e] * inv + (be[e] - mean * inv);
      }
      const uint64_t e0 = (uint64_t)(row0 + tok) * HID + (uint64_t)(32 * ks + 8 * g);
      const f32x4 lo = b4r_drop4(dce, (f32x4){y[0], y[1], y[2], y[3]}, e0), hi = b4r_drop4(dce, (f32x4){y[4], y[5], y[6], y[7]}, e0 + 4);
      xv[ks] = cat(lo, hi);
    }
    emb_mean = mean; emb_rstd = rstd;   // x and its statistics are stored behind the staging barrier (which would wait for the stores)
    xsrc = p.x_out;
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xv[ks] = load8(p.x + (row0 + tokc) * HID + 32 * ks + 8 * g);   // in flight during the staging
  }
  stage_weight_pair(big, p.Wqkv, HID, 3 * HID, woimg, p.Wo, HID, HID, nthreads);
  if ((int)threadIdx.x < KTE * 16) sAdd[threadIdx.x] = (int)threadIdx.x < L ? (1.0f - (float)mval) * -1e9f : -INFINITY;
  if (bias_one_turn) { if (threadIdx.x < 3 * HID) sbq[threadIdx.x] = bq_val; }
  else for (int k = threadIdx.x; k < 3 * HID; k += nthreads) sbq[k] = p.bqkv[k];
  bf16x8 xh[2], xl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) split8(xv[ks], xh[ks], xl[ks]);
  AF_MARK(1);
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
  AF_MARK(2);
  if (p.ids != nullptr && tok < L) {   // the embedding stage's outputs (their acknowledgements arrive during the QKV phase)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      *reinterpret_cast<f32x4*>(p.x_out + (row0 + tok) * HID + 32 * ks + 8 * g) = (f32x4){xv[ks][0], xv[ks][1], xv[ks][2], xv[ks][3]};
      *reinterpret_cast<f32x4*>(p.x_out + (row0 + tok) * HID + 32 * ks + 8 * g + 4) = (f32x4){xv[ks][4], xv[ks][5], xv[ks][6], xv[ks][7]};
    }
    if (g == 0) {
      if (p.mean0) p.mean0[row0 + tok] = emb_mean;
      if (p.rstd0) p.rstd0[row0 + tok] = emb_rstd;
    }
  }

  // ---- phase 1: q, k, v of this wave's 16 tokens.  Tile a of feature block fb holds features 32 fb + 8p + 4a + e on its
  // rows 4p + e, so that a stacked pair is a B operand in natural feature order (b4r_ffn_rx.hip) ---------------------------
  const int tr_w[2][2] = {{sub_off(8 * (g & 1) + qq, pp), sub_off(8 * (g & 1) + 4 + qq, pp)},
                          {sub_off(8 * (g & 1) + qq, pp) + 8, sub_off(8 * (g & 1) + 4 + qq, pp) + 8}};
  f32x4 qkv[6][2];
#pragma unroll
  for (int fb = 0; fb < 6; ++fb) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f32x4 c = *reinterpret_cast<const f32x4*>(&sbq[32 * fb + 8 * g + 4 * a]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char* t = big + sub_base(2 * ks + (g >> 1), fb, 6);
        c = mfma3(tr_pair(t + tr_w[a][0], t + tr_w[a][1]), tr_pair(t + SUB + tr_w[a][0], t + SUB + tr_w[a][1]), xh[ks], xl[ks], c);
      }
      if (fb < 2) c = c * p.qscale;
      qkv[fb][a] = c;
    }
  }
  const bool live = tok < L;
  if (p.qkv && live) {
    float* dst = p.qkv + (row0 + tok) * (3 * HID) + 8 * g;
#pragma unroll
    for (int fb = 0; fb < 6; ++fb) {
      *reinterpret_cast<f32x4*>(dst + 32 * fb) = qkv[fb][0];
      *reinterpret_cast<f32x4*>(dst + 32 * fb + 4) = qkv[fb][1];
    }
  }
  bf16x8 qh[2], ql[2];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) split8(cat(qkv[hd][0], qkv[hd][1]), qh[hd], ql[hd]);
  AF_MARK(3);
  __syncthreads();   // every wave is done with the Wqkv image: the K / V images may overwrite it
  AF_MARK(4);

  // ---- phase 2: this wave's rows of the K / V images (zero rows for pad tokens), and the zero tiles beyond KT ----------
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    char* img = big + hd * KTE * TILE_BYTES;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      bf16x4 kh, kl, vh, vl;
      b4r_split4(live ? qkv[2 + hd][a] : z, kh, kl);
      b4r_split4(live ? qkv[4 + hd][a] : z, vh, vl);
      char* dst = img + img_off(16 * wave + i, g) + 8 * a;
      *reinterpret_cast<bf16x4*>(dst) = kh;
      *reinterpret_cast<bf16x4*>(dst + IMG_BYTES) = kl;
      *reinterpret_cast<bf16x4*>(dst + 2 * IMG_BYTES) = vh;
      *reinterpret_cast<bf16x4*>(dst + 3 * IMG_BYTES) = vl;
    }
  }
  {
    const int kt_live = (L + 15) >> 4;   // = number of waves
    for (int f = threadIdx.x; f < (KTE - kt_live) * 2 * (TILE_BYTES / 16); f += nthreads) {
      const int per = (KTE - kt_live) * (TILE_BYTES / 16);
      const int hd = f / per, r = f - hd * per;
      *reinterpret_cast<f32x4*>(big + hd * KTE * TILE_BYTES + kt_live * TILE_BYTES + 16 * r) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  AF_MARK(5);
  __syncthreads();
  AF_MARK(6);

  // ---- phase 3: attention per head (attn_rx_fwd_kernel's body) --------------------------------------------------------------
  const FragAddr fa = frag_addr(lane);
  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  f32x4 o[2][2];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    const char* img = big + hd * KTE * TILE_BYTES;
    AF_MARK(7 + hd);
    f32x4 acc[KTE];
#pragma unroll
    for (int t = 0; t < KTE; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const char* tile = img + fa.row + TILE_BYTES * t;
      if (t < KTT) acc[t] = mfma3(row_frag<0>(tile), row_frag<1>(tile), qh[hd], ql[hd], acc[t]);   // S^T = K.Q^T
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < KTT; ++t) {
      const f32x4 ad = *reinterpret_cast<const f32x4*>(&sAdd[16 * t + 4 * g]);
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[t][r] += ad[r]; m = fmaxf(m, acc[t][r]); }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    f32x4 sum4v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KTT; ++t) {
      const f32x4 d = (acc[t] - m) * 1.4426950408889634f;   // (score - max) first: see attn_rx_fwd_kernel
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_amdgcn_exp2f(d[r]);
      sum4v += acc[t];
    }
    const float sum = quad_sum(sum4(sum4v));
    const float inv = 1.0f / sum;
    const int64_t bh = (int64_t)b * 2 + hd;
    if (g == 0 && live && p.lse) p.lse[bh * L + tok] = (m - amax) + __logf(sum);
    if (dcp.on) {
      const uint64_t dbase = ((uint64_t)bh * L + (uint64_t)(live ? tok : 0)) * (uint64_t)B4R_ATTN_PITCH;
      uint32_t w[2] = {0u, 0u};
#pragma unroll
      for (int t = 0; t < KTT; ++t) {
        const B4rKeep4 k4 = b4r_keep4p(dcp, dbase + (uint64_t)(16 * t + 4 * g));
        const f32x4 ps = acc[t] * (inv * dcp.scale);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t][s] = k4.k[s] ? ps[s] : 0.f;
        w[t >> 3] |= k4.bits() << (4 * (t & 7));
      }
      if (p.bits) {
        uint32_t* wo = p.bits + ((bh * p.KT + wave) * 2) * 64 + lane;
        wo[0] = w[0];
        wo[64] = w[1];
      }
      if (p.bits32) {   // [b][head][key tile T][query tile][16 register pairs][2]: bit k of the word = key 32 T + k
        const int NT = (L + 31) >> 5, qt = tok >> 5, qr = tok & 31;
        const int slot = (qr & 24) | ((qr & 3) << 1) | ((qr >> 2) & 1);   // query 16s + 8a + 4h' + b -> 16s + 8a + 2b + h'
        for (int T = 0; T < NT; ++T) {
          const uint32_t by = (w[T >> 2] >> (8 * (T & 3))) & 0xFFu;        // the nibbles of the 16-key tiles 2T, 2T + 1
          uint32_t part = ((by & 15u) | ((by >> 4) << 16)) << (4 * g);     // rows 4g .. 4g+3 of each
          part |= (uint32_t)__shfl_xor((int)part, 16, 64);
          part |= (uint32_t)__shfl_xor((int)part, 32, 64);
          if (g == 0 && live) p.bits32[((bh * NT + T) * NT + qt) * 32 + slot] = part;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < KTT; ++t) acc[t] = acc[t] * inv;
    }
    o[hd][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    o[hd][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < KTE / 2; ++tp) {   // O^T[dd][query] += V^T[dd][keys of two tiles] . P^T[keys][query]
      bf16x8 ph, pl;
      split8(cat(acc[2 * tp], acc[2 * tp + 1]), ph, pl);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const char* tile = img + fa.tr[db] + TILE_BYTES * 2 * tp;
        o[hd][db] = mfma3(tr_frag<2>(tile), tr_frag<3>(tile), ph, pl, o[hd][db]);
      }
    }
  }
  AF_MARK(9);
  if (live && p.ctx) {
    float* dst = p.ctx + (row0 + tok) * HID + 4 * g;
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
      *reinterpret_cast<f32x4*>(dst + 32 * hd) = o[hd][0];
      *reinterpret_cast<f32x4*>(dst + 32 * hd + 16) = o[hd][1];
    }
  }

  // ---- phase 4: y^T = Wo^T.ctx^T, bias, dropout, residual, LayerNorm -----------------------------------------------------------
  // k-slot (g, j) of head hd's stacked pair = context column 32 hd + 16 (j >> 2) + 4g + (j & 3): rows 4g .. of row tile 2 hd
  // of the Wo image for j < 4, of row tile 2 hd + 1 for j >= 4
  f32x4 y[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) y[hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    bf16x8 ch, cl;
    split8(cat(o[hd][0], o[hd][1]), ch, cl);
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      const char* t0 = woimg + sub_base(2 * hd, hb >> 1, 2) + fa.tr[hb & 1];
      const char* t1 = t0 + 2 * 2 * SUB;
      y[hb] = mfma3(tr_pair(t0, t1), tr_pair(t0 + SUB, t1 + SUB), ch, cl, y[hb]);
    }
  }
  AF_MARK(10);
  const DropCtx dco = b4r_drop_ctx(p.drop_o);
  f32x4 z[4];
  float s = 0.f;
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const f32x4 v = y[hb] + *reinterpret_cast<const f32x4*>(p.bo + 16 * hb + 4 * g);
    const f32x4 res = *reinterpret_cast<const f32x4*>(xsrc + (row0 + tokc) * HID + 16 * hb + 4 * g);
    z[hb] = res + b4r_drop4(dco, v, (uint64_t)(row0 + tok) * HID + (uint64_t)(16 * hb + 4 * g));
    s += sum4(z[hb]);
  }
  const float mean = quad_sum(s) * (1.0f / HID);
  float q = 0.f;
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const f32x4 d = z[hb] - mean;
    q += sum4(d * d);
  }
  const float rstd = rsqrtf(quad_sum(q) * (1.0f / HID) + p.eps);
  AF_MARK(11);
  if (live) {
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      const int64_t off = (row0 + tok) * HID + 16 * hb + 4 * g;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(p.g1 + 16 * hb + 4 * g);
      const f32x4 be = *reinterpret_cast<const f32x4*>(p.be1 + 16 * hb + 4 * g);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float inv = rstd * gm[e];
        out[e] = fmaf(z[hb][e], inv, fmaf(-mean, inv, be[e]));
      }
      if (p.z1) *reinterpret_cast<f32x4*>(p.z1 + off) = z[hb];
      if (p.x1) *reinterpret_cast<f32x4*>(p.x1 + off) = out;
    }
    if (g == 0) {
      if (p.mean1) p.mean1[row0 + tok] = mean;
      if (p.rstd1) p.rstd1[row0 + tok] = rstd;
    }
  }
  AF_MARK(12);
}

// -----------------------------------------------------------------------------------------------------------
// backward.  One workgroup per sequence, wave w owns token tile w as queries AND as keys; every 16 x 16 block of the score
// matrix is formed ONCE (round 1 formed it twice: in the dQ kernel by its query rows, in the dK/dV kernel by its key rows,
// with all of the exp / dropout / hi-lo split work around it):
//   step s, wave w:  key tile t = (w + s) mod KT
//     S^T = K_t.Q_w^T, dA^T = V_t.dO_w^T (K = 32 features)        -> pr, dropped pr, dS in registers (keys on rows, queries on lanes)
//     dQ_w^T += K_t^T.dS^T        (sums over the block's 16 keys: the accumulator tile is the B operand, v_mfma_f32_16x16x16_bf16)
//     dK_t^T += Q_w^T.dS, dV_t^T += dO_w^T.Pd   (sum over the block's 16 queries: dS / Pd transposed through a wave-private LDS
//                                tile, 8-byte writes + ds_read_b64_tr_b16), added into fp32 accumulators of key tile t in LDS
//   In one step every wave works on a different key tile, a barrier separates the steps: the accumulation order of a key
//   tile is fixed (waves t, t-1, t-2, ...), so dK / dV are bitwise reproducible without atomics.
// Per head: q, k, v are RECOMPUTED from x (the forward does not have to store qkv: 39 MB per layer), dctx = dropmask(dz1).Wo^T is
// formed in registers, K / V rows go to the attention images; the Q / dO rows of a wave are only ever read by that wave (as
// transposed A operands), their image space becomes the dK / dV accumulators.  After both heads:
//   dX^T = Wqkv.dqkv^T + dz1^T, then the backward of the LayerNorm that produced x (the previous layer's output LayerNorm, or
//   for layer 0 the embedding stage: dropout, then LayerNorm of table[id] + position) -> `da`, gamma / beta partial sums.
// dqkv [N, 3H] is written for the weight-gradient product dWqkv = x^T.dqkv (b4r_gemm_tn_f32), the only consumer left.
// LDS: [tile][K hi, K lo, V hi, V lo | Q hi, Q lo, dO hi, dO lo -> dK^T, dV^T accumulators] 8 KB per 16 tokens (the weight
// images overlay it between the heads), 2 KB of transposition scratch per wave, the key mask, biases.
// -----------------------------------------------------------------------------------------------------------
struct AbBwdP {
  const float* x; const float* dz1; const float* ctx; const float* lse; const uint32_t* bits; const int64_t* mask;
  const float* Wqkv; const float* bqkv; const float* Wo;
  const float* zprev; const float* meanp; const float* rstdp; const float* gprev;     // the LayerNorm that produced x
  const int64_t* ids; const float* table; const float* pos; int V;                    // ... or the embedding stage (ids != NULL)
  float* dqkv; float* da; float* ln_part;
  int B, L, KT;
  float qscale;
  DropArgs drop_p, drop_o, drop_e;
};

// timing experiments only (tools/build_variant.sh): 1 = no sweep, 2 = no accumulator read-modify-write, 4 = no barrier per step,
// 8 = one weight staging instead of three
#ifndef AB_EXP
#define AB_EXP 0
#endif
constexpr int BT = 8 * 1024;   // bytes of one token tile of the backward's LDS region
constexpr int BX = 4 * IMG_BYTES;   // offset of the Q / dO images (later the accumulators) inside a tile

typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(const bf16x4 a, const bf16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16x3(const bf16x4 ah, const bf16x4 al, const bf16x4 bh, const bf16x4 bl, f32x4 c) {
  c = mfma16(al, bh, c);
  c = mfma16(ah, bl, c);
  c = mfma16(ah, bh, c);
  return c;
}
__device__ __forceinline__ bf16x4 tr_one(const char* a) {
  return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a));
}
__device__ __forceinline__ bf16x8 row16(const char* a) { return *reinterpret_cast<const bf16x8*>(a); }

template <bool EMBED>
__global__ __launch_bounds__(1024) void attn_block_bwd_kernel(AbBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_ab[];
  const int KT = p.KT, L = p.L;
  const int rbytes = KT * BT > 64 * 1024 ? KT * BT : 64 * 1024;
  char* R = smem_ab;                                   // tiles; between the heads: [Wqkv image 48 KB | Wo image 16 KB]
  char* scratch_all = smem_ab + rbytes;                // 2 KB per wave
  float* sAdd = reinterpret_cast<float*>(scratch_all + KT * 2048);   // [KT * 16]
  float* sbq = sAdd + KT * 16;                         // bqkv [192]
  float* sred = sbq + 3 * HID;                         // [KT][128] LayerNorm partials (end of the kernel)

  const int nthreads = blockDim.x;
  const int b = blockIdx.x;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4, qq = i >> 2, pp = i & 3;
  const int tok = 16 * wave + i, tokc = min(tok, L - 1);
  const bool live = tok < L;
  char* scratch = scratch_all + wave * 2048;
  char* woimg = R + 48 * 1024;
  // b4r_seq_amax, its barrier folded into the first one of the head loop
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;   // one turn: at least 4 threads per token
  const int any_key = mval != 0 ? 1 : 0;
  float amax = 0.0f;

  if ((int)threadIdx.x < KT * 16) sAdd[threadIdx.x] = (int)threadIdx.x < L ? (1.0f - (float)mval) * -1e9f : -INFINITY;
  for (int k = threadIdx.x; k < 3 * HID; k += nthreads) sbq[k] = p.bqkv[k];

  // lane constants.  tr_w: transposed fragment whose 16 columns are the interleaved features 8p' + 4a + e (b4r_ffn_rx.hip);
  // rows = a natural 32-deep k block (weights) or one 16-row tile (images)
  const int trw_w[2][2] = {{sub_off(8 * (g & 1) + qq, pp), sub_off(8 * (g & 1) + 4 + qq, pp)},
                           {sub_off(8 * (g & 1) + qq, pp) + 8, sub_off(8 * (g & 1) + 4 + qq, pp) + 8}};
  const int trw_t[2] = {sub_off(4 * g + qq, pp), sub_off(4 * g + qq, pp) + 8};   // rows 4g .. 4g+3 of a tile, features 8p'+4a+e
  const int row_i = sub_off(i, g);                                               // row i, columns 8g .. 8g+7
  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  const float pscale = dcp.on ? dcp.scale : 1.0f;

  f32x4 dxacc[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) dxacc[hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 gq[2], gk[2], gv[2];   // dq (scaled), dk, dv of the head just finished: own tokens, interleaved feature tiles

  AB_MARK(0);
  for (int hd = 0; hd <= 2; ++hd) {
    // the tile region is free (previous head done; first pass: sAdd / sbq written)
    if (hd == 0) amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
    else __syncthreads();
    AB_MARK(1 + 10 * hd);
    if (!(AB_EXP & 8) || hd == 0) {
    if (hd < 2) stage_weight_pair(R, p.Wqkv, HID, 3 * HID, woimg, p.Wo, HID, HID, nthreads);
    else stage_weight(R, p.Wqkv, HID, 3 * HID, nthreads);
    }
    __syncthreads();
    AB_MARK(2 + 10 * hd);
    if (hd > 0) {
      // dX^T[16 hb + ..][token] += Wqkv[.., features of head hd-1] . dqkv^T: A = rows of the Wqkv image (natural k order)
      const int ph = hd - 1;
      // dqkv of that head leaves here, BEHIND the two barriers of the staging: a barrier waits for the wave's outstanding stores, and
      // right after the sweep that put 2.7 us of store acknowledgements on the critical path of every head
      if (live) {   // (keeping the last head's until the very end of the kernel was tried: 52 spilled registers, slower)
        float* dst = p.dqkv + (row0 + tok) * (3 * HID) + 32 * ph + 8 * g;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          *reinterpret_cast<f32x4*>(dst + 4 * a) = gq[a];
          *reinterpret_cast<f32x4*>(dst + HID + 4 * a) = gk[a];
          *reinterpret_cast<f32x4*>(dst + 2 * HID + 4 * a) = gv[a];
        }
      }
      bf16x8 bh_[3], bl_[3];
      split8(cat(gq[0], gq[1]), bh_[0], bl_[0]);
      split8(cat(gk[0], gk[1]), bh_[1], bl_[1]);
      split8(cat(gv[0], gv[1]), bh_[2], bl_[2]);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int fb = 2 * j + ph;
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
          const char* s0 = R + sub_base(hb, fb, 6) + row_i;
          dxacc[hb] = mfma3(row16(s0), row16(s0 + SUB), bh_[j], bl_[j], dxacc[hb]);
        }
      }
    }
    AB_MARK(3 + 10 * hd);
    if (hd == 2) break;

    // ---- q, k, v of this head for the wave's tokens (recomputed) and dctx = dropmask(dz1).Wo^T -------------------------
    f32x4 qt[2], kt[2], vt[2], dct[2];
    {
      bf16x8 xh[2], xl[2], yh[2], yl[2];
      const DropCtx dco = b4r_drop_ctx(p.drop_o);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        split8(load8(p.x + (row0 + tokc) * HID + 32 * ks + 8 * g), xh[ks], xl[ks]);
        f32x8 dy = load8(p.dz1 + (row0 + tokc) * HID + 32 * ks + 8 * g);
        if (dco.on) {
          const uint64_t e0 = (uint64_t)(row0 + tok) * HID + (uint64_t)(32 * ks + 8 * g);
          dy = cat(b4r_drop4(dco, lo4(dy), e0), b4r_drop4(dco, hi4(dy), e0 + 4));
        }
        split8(dy, yh[ks], yl[ks]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int fb = 2 * j + hd;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          f32x4 c = *reinterpret_cast<const f32x4*>(&sbq[32 * fb + 8 * g + 4 * a]);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const char* t = R + sub_base(2 * ks + (g >> 1), fb, 6);
            c = mfma3(tr_pair(t + trw_w[a][0], t + trw_w[a][1]), tr_pair(t + SUB + trw_w[a][0], t + SUB + trw_w[a][1]), xh[ks], xl[ks], c);
          }
          if (j == 0) qt[a] = c * p.qscale; else if (j == 1) kt[a] = c; else vt[a] = c;
        }
      }
      // dctx^T tile a: rows 4p + e = context columns 32 hd + 8p + 4a + e  =  rows of the Wo image (natural k = output column)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int wrow = 32 * hd + 8 * qq + 4 * a + pp;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const char* s0 = woimg + sub_base(wrow >> 4, ks, 2) + sub_off(wrow & 15, g);
          c = mfma3(row16(s0), row16(s0 + SUB), yh[ks], yl[ks], c);
        }
        dct[a] = c;
      }
    }
    AB_MARK(4 + 10 * hd);
    // D = sum_c dctx * ctx over this head's 32 columns (the softmax backward's row term)
    float Dq;
    {
      const float* cr = p.ctx + (row0 + tokc) * HID + 32 * hd + 8 * g;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr), c1 = *reinterpret_cast<const f32x4*>(cr + 4);
      Dq = quad_sum(sum4(dct[0] * c0) + sum4(dct[1] * c1));
    }
    const int64_t bh = (int64_t)b * 2 + hd;
    const float lse_q = live ? p.lse[bh * L + tok] : INFINITY;   // +inf: probability 0 for pad queries
    uint32_t wbits[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
    if (dcp.on) {
      const uint32_t* wi = p.bits + ((bh * KT + wave) * 2) * 64 + lane;
      wbits[0] = wi[0];
      wbits[1] = wi[64];
    }
    __syncthreads();   // every wave is done with the weight images
    AB_MARK(5 + 10 * hd);

    // ---- own rows of the K, V, Q, dO images ------------------------------------------------------------------------------
    char* mytile = R + wave * BT;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      bf16x4 h, l;
      char* dst = mytile + sub_off(i, g) + 8 * a;
      b4r_split4(kt[a], h, l);
      *reinterpret_cast<bf16x4*>(dst) = h; *reinterpret_cast<bf16x4*>(dst + IMG_BYTES) = l;
      b4r_split4(vt[a], h, l);
      *reinterpret_cast<bf16x4*>(dst + 2 * IMG_BYTES) = h; *reinterpret_cast<bf16x4*>(dst + 3 * IMG_BYTES) = l;
      b4r_split4(qt[a], h, l);
      *reinterpret_cast<bf16x4*>(dst + BX) = h; *reinterpret_cast<bf16x4*>(dst + BX + IMG_BYTES) = l;
      b4r_split4(dct[a], h, l);
      *reinterpret_cast<bf16x4*>(dst + BX + 2 * IMG_BYTES) = h; *reinterpret_cast<bf16x4*>(dst + BX + 3 * IMG_BYTES) = l;
    }
    // the wave reads back only what it wrote itself: LDS operations of one wave are ordered
    bf16x8 qh, ql, doh, dol;       // B operands of S^T = K.Q^T and dA^T = V.dO^T (k = feature, natural order)
    bf16x4 qT[2][2], dT[2][2];     // [a][hi, lo]: Q^T / dO^T as A operands (row = feature 8p+4a+e, k = query 4g+j)
    split8(cat(qt[0], qt[1]), qh, ql);
    split8(cat(dct[0], dct[1]), doh, dol);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      qT[a][0] = tr_one(mytile + BX + trw_t[a]);
      qT[a][1] = tr_one(mytile + BX + IMG_BYTES + trw_t[a]);
      dT[a][0] = tr_one(mytile + BX + 2 * IMG_BYTES + trw_t[a]);
      dT[a][1] = tr_one(mytile + BX + 3 * IMG_BYTES + trw_t[a]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the transposed reads have landed before the space is zeroed
    __builtin_amdgcn_sched_barrier(0);
    // the Q / dO space of this wave's tile becomes its dK^T / dV^T accumulators: [dk a0 | dk a1 | dv a0 | dv a1], 1 KB each
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(mytile + BX + j * 1024 + lane * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();   // K / V rows of every tile are in place, every accumulator is zero
    AB_MARK(6 + 10 * hd);

    // ---- the sweep: one 16 x 16 block of the score matrix per step ------------------------------------------------------------
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int s = 0; s < ((AB_EXP & 1) ? 0 : KT); ++s) {
      int t = wave + s;
      if (t >= KT) t -= KT;
      const char* tile = R + t * BT;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 sc = mfma3(row16(tile + row_i), row16(tile + IMG_BYTES + row_i), qh, ql, z);                       // S^T
      const f32x4 dA = mfma3(row16(tile + 2 * IMG_BYTES + row_i), row16(tile + 3 * IMG_BYTES + row_i), doh, dol, z);   // dA^T
      const f32x4 ad = *reinterpret_cast<const f32x4*>(&sAdd[16 * t + 4 * g]);
      const uint32_t nib = wbits[(t >> 3) & 1] >> (4 * (t & 7));
      f32x4 ds, pd;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __expf(((sc[r] + ad[r]) - amax) - lse_q);
        const bool keep = (nib >> r) & 1u;
        pd[r] = keep ? pr * pscale : 0.f;
        const float da_ = keep ? dA[r] * pscale : 0.f;
        ds[r] = pr * (da_ - Dq);
      }
      bf16x4 dsh, dsl, pdh, pdl;
      b4r_split4(ds, dsh, dsl);
      b4r_split4(pd, pdh, pdl);
      // dQ^T[feature][query] += K_t^T[feature][keys] . dS^T[keys][query]
#pragma unroll
      for (int a = 0; a < 2; ++a)
        dq[a] = mfma16x3(tr_one(tile + trw_t[a]), tr_one(tile + IMG_BYTES + trw_t[a]), dsh, dsl, dq[a]);
      // transpose dS^T, Pd^T (keys on rows, queries on lanes) -> [query][key] tiles, read back with queries on the k slots
      {
        char* w = scratch + i * 32 + g * 8;
        *reinterpret_cast<bf16x4*>(w) = dsh; *reinterpret_cast<bf16x4*>(w + 512) = dsl;
        *reinterpret_cast<bf16x4*>(w + 1024) = pdh; *reinterpret_cast<bf16x4*>(w + 1536) = pdl;
      }
      const char* rd = scratch + (4 * g + qq) * 32 + pp * 8;
      const bf16x4 dsTh = tr_one(rd), dsTl = tr_one(rd + 512), pdTh = tr_one(rd + 1024), pdTl = tr_one(rd + 1536);
      // dK_t^T[feature][key] += Q_w^T[feature][queries] . dS[queries][key] ;  dV_t^T += dO_w^T . Pd
      // (LDS-side adds, ds_add_f32, were measured here: 12x slower than this read-modify-write)
      char* acc = R + t * BT + BX + lane * 16;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const f32x4 dkp = mfma16x3(qT[a][0], qT[a][1], dsTh, dsTl, z);
        const f32x4 dvp = mfma16x3(dT[a][0], dT[a][1], pdTh, pdTl, z);
        f32x4* ak = reinterpret_cast<f32x4*>(acc + a * 1024);
        f32x4* av = reinterpret_cast<f32x4*>(acc + (2 + a) * 1024);
        if (AB_EXP & 2) { asm volatile("" :: "v"(dkp), "v"(dvp)); continue; }
        *ak = *ak + dkp;
        *av = *av + dvp;
      }
      if (!(AB_EXP & 4)) __syncthreads();   // the next step adds into other tiles; fixed order of the additions into each tile
    }
    AB_MARK(7 + 10 * hd);
    // results of this head for the wave's tokens
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      gq[a] = dq[a] * p.qscale;
      gk[a] = *reinterpret_cast<const f32x4*>(mytile + BX + a * 1024 + lane * 16);
      gv[a] = *reinterpret_cast<const f32x4*>(mytile + BX + (2 + a) * 1024 + lane * 16);
    }
  }

  AB_MARK(30);
  // ---- dx = dX + dz1 (residual), then back through the LayerNorm (and, for layer 0, the dropout) that produced x -------------
  const DropCtx dce = b4r_drop_ctx(p.drop_e);
  const float mean = p.meanp[row0 + tokc], rstd = p.rstdp[row0 + tokc];
  f32x4 ge[4], xhat[4], dgam[4], dbet[4];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const int col = 16 * hb + 4 * g;
    f32x4 dx = dxacc[hb] + *reinterpret_cast<const f32x4*>(p.dz1 + (row0 + tokc) * HID + col);
    f32x4 zz;
    if (EMBED) {
      dx = b4r_drop4(dce, dx, (uint64_t)(row0 + tok) * HID + (uint64_t)col);
      int64_t id = p.ids[row0 + tokc];
      if (id < 0 || id >= p.V) id = 0;   // as the forward: out-of-range ids read the PAD row
      zz = *reinterpret_cast<const f32x4*>(p.table + id * HID + col) + *reinterpret_cast<const f32x4*>(p.pos + (int64_t)tokc * HID + col);
    } else {
      zz = *reinterpret_cast<const f32x4*>(p.zprev + (row0 + tokc) * HID + col);
    }
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gprev + col);
    xhat[hb] = (zz - mean) * rstd;
    ge[hb] = dx * gm;
    s1 += sum4(ge[hb]);
    s2 += sum4(ge[hb] * xhat[hb]);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    dgam[hb] = live ? dx * xhat[hb] : zero;
    dbet[hb] = live ? dx : zero;
  }
  const float c1 = quad_sum(s1) * (1.0f / HID), c2 = quad_sum(s2) * (1.0f / HID);
  AB_MARK(32);
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {   // column sums over the wave's 16 tokens
#pragma unroll
    for (int e = 0; e < 4; ++e) { dgam[hb][e] = row_sum15(dgam[hb][e]); dbet[hb][e] = row_sum15(dbet[hb][e]); }
    if (i == 15) {
      *reinterpret_cast<f32x4*>(&sred[wave * 128 + 16 * hb + 4 * g]) = dgam[hb];
      *reinterpret_cast<f32x4*>(&sred[wave * 128 + 64 + 16 * hb + 4 * g]) = dbet[hb];
    }
  }
  AB_MARK(34);
  __syncthreads();   // before the wave's big stores: a barrier behind them would wait for their acknowledgements
  AB_MARK(35);
  for (int k = threadIdx.x; k < 128; k += nthreads) {   // a workgroup may be a single wave (L <= 16)
    float r = 0.f;
    for (int w = 0; w < KT; ++w) r += sred[w * 128 + k];
    p.ln_part[(int64_t)b * 128 + k] = r;
  }
  if (live) {
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      f32x4 dz;
#pragma unroll
      for (int e = 0; e < 4; ++e) dz[e] = rstd * (ge[hb][e] - c1 - xhat[hb][e] * c2);
      *reinterpret_cast<f32x4*>(p.da + (row0 + tok) * HID + 16 * hb + 4 * g) = dz;
    }
  }
  AB_MARK(33);
  AB_MARK(31);
}

size_t bwd_lds(int KT) {
  const size_t r = (size_t)KT * BT > 64 * 1024 ? (size_t)KT * BT : 64 * 1024;
  return r + (size_t)KT * 2048 + ((size_t)KT * 16 + 3 * HID + (size_t)KT * 128) * sizeof(float);
}

template <int KTT>
size_t fwd_lds() {
  constexpr int KTE = (KTT + 1) & ~1;
  constexpr int KV_BYTES = 2 * KTE * TILE_BYTES;
  constexpr int BIG = KV_BYTES > 48 * 1024 ? KV_BYTES : 48 * 1024;
  return 16 * 1024 + BIG + (KTE * 16 + 3 * HID) * sizeof(float);
}

bool al16(const void* q) { return q == nullptr || b4r_aligned16(q); }

}  // namespace

int32_t b4r_attn32_supported(int32_t hidden_size, int32_t num_heads, int32_t L);
int32_t b4r_attn32_preferred(int32_t hidden_size, int32_t num_heads, int32_t L);   // supported AND long enough to pay (b4r_attn32.hip)
int b4r_attn32_bwd(const b4r_attn_block_bwd_desc* d, b4r_stream_t stream);
int64_t b4r_attn_rx_keep_words(int B, int L, int heads);
static bool use_attn32() {
  static const bool on = !(getenv("B4R_ATTN32") && atoi(getenv("B4R_ATTN32")) == 0);
  return on;
}

bool b4r_attn32_active(int H, int heads, int L) { return use_attn32() && b4r_attn32_preferred(H, heads, L) != 0; }

extern "C" int32_t b4r_attn_block_supported(int32_t hidden_size, int32_t num_heads, int32_t L) {
  return (hidden_size == HID && num_heads == 2 && L > 0 && L <= 256 && b4r_get_gemm_mode() == B4R_GEMM_BF16X3) ? 1 : 0;
}

extern "C" int32_t b4r_attn_block_bwd_supported(int32_t hidden_size, int32_t num_heads, int32_t L) {
  return (b4r_attn_block_supported(hidden_size, num_heads, L) && L <= 208) ? 1 : 0;   // 10 KB of LDS per 16 tokens
}
#ifdef AB_PROF
extern "C" int b4r_debug_af_prof(long long* host_out) {   // 16 stamps of the last forward launch
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_af_prof), 16 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
extern "C" int b4r_debug_ab_prof(long long* host_out) {   // 64 stamps of the last backward launch (after a device synchronisation)
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ab_prof), 64 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
#endif
extern "C" int64_t b4r_attn_block_bwd_scratch_floats(int32_t B) { return (int64_t)(B > 0 ? B : 0) * 128; }
extern "C" int64_t b4r_attn_block_bwd_dw_scratch_floats(int32_t B) {
  return (int64_t)(B > 0 ? B : 0) * (HID * 3 * HID + 3 * HID + HID * HID + HID);   // dWqkv, dbqkv, dWo, dbo partials per sequence
}

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);

extern "C" int b4r_attn_block_bwd(const b4r_attn_block_bwd_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_attn_block_bwd: null descriptor");
  // the 32-token-tile kernel where it is preferred, and wherever the descriptor asks for what only it does (weight gradients inside
  // the launch, sparse dz1); the forward of the same step wrote the dropout decisions in both layouts (bits32 below)
  if (use_attn32() && b4r_attn32_supported(d->H, d->heads, d->L) &&
      (b4r_attn32_preferred(d->H, d->heads, d->L) || d->dWqkv != nullptr || d->dz1_slot_positions != nullptr || d->dqkv == nullptr))
    return b4r_attn32_bwd(d, stream);
  B4R_CHECK_ARG(d->dWqkv == nullptr && d->dqkv != nullptr, B4R_E_SHAPE, "b4r_attn_block_bwd: dWqkv inside the launch needs L <= 224 (this path writes dqkv)");
  B4R_CHECK_ARG(b4r_attn_block_bwd_supported(d->H, d->heads, d->L), B4R_E_SHAPE,
                "b4r_attn_block_bwd: needs hidden size 64, 2 heads, L <= 208 and the bf16x3 mode (H=%d heads=%d L=%d)", d->H, d->heads,
                d->L);
  B4R_CHECK_ARG(d->B > 0 && d->x && d->dz1 && d->ctx && d->lse && d->input_mask && d->Wqkv && d->bqkv && d->Wo && d->prev_mean &&
                    d->prev_rstd && d->prev_gamma && d->dqkv && d->dx_prev && d->dprev_gamma && d->scratch,
                B4R_E_BADARG, "b4r_attn_block_bwd: null argument");
  const bool embed = d->emb_ids != nullptr;
  B4R_CHECK_ARG(embed ? (d->emb_table && d->emb_pos && d->emb_vocab > 0) : (d->prev_z != nullptr), B4R_E_BADARG,
                "b4r_attn_block_bwd: needs prev_z, or emb_ids + emb_table + emb_pos");
  B4R_CHECK_ARG(al16(d->x) && al16(d->dz1) && al16(d->ctx) && al16(d->Wqkv) && al16(d->Wo) && al16(d->prev_z) && al16(d->prev_gamma) &&
                    al16(d->emb_table) && al16(d->emb_pos) && al16(d->dqkv) && al16(d->dx_prev) && al16(d->keep_bits),
                B4R_E_ALIGN, "b4r_attn_block_bwd: operands must be 16-byte aligned");
  AbBwdP p{};
  p.x = d->x; p.dz1 = d->dz1; p.ctx = d->ctx; p.lse = d->lse; p.bits = d->keep_bits; p.mask = d->input_mask;
  p.Wqkv = d->Wqkv; p.bqkv = d->bqkv; p.Wo = d->Wo;
  p.zprev = d->prev_z; p.meanp = d->prev_mean; p.rstdp = d->prev_rstd; p.gprev = d->prev_gamma;
  p.ids = d->emb_ids; p.table = d->emb_table; p.pos = d->emb_pos; p.V = d->emb_vocab;
  p.dqkv = d->dqkv; p.da = d->dx_prev; p.ln_part = d->scratch;
  p.B = d->B; p.L = d->L; p.KT = b4r_cdiv(d->L, 16);
  p.qscale = 1.0f / sqrtf(32.0f);
  p.drop_p = b4r_make_drop(d->rng, d->probs_stream, d->probs_rate, d->rng != nullptr);
  p.drop_o = b4r_make_drop(d->rng, d->out_stream, d->out_rate, d->rng != nullptr);
  p.drop_e = b4r_make_drop(d->rng, d->emb_stream, d->emb_rate, d->rng != nullptr && embed);
  B4R_CHECK_ARG(!p.drop_p.rng || d->keep_bits, B4R_E_BADARG, "b4r_attn_block_bwd: attention dropout needs the forward's keep_bits");
  const size_t sh = bwd_lds(p.KT);
  const dim3 grid((unsigned)d->B), block((unsigned)(64 * p.KT));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (embed) {
    rc = b4r_raise_lds((const void*)attn_block_bwd_kernel<true>, sh, "b4r_attn_block_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL((attn_block_bwd_kernel<true>), grid, block, sh, s, p);
  } else {
    rc = b4r_raise_lds((const void*)attn_block_bwd_kernel<false>, sh, "b4r_attn_block_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL((attn_block_bwd_kernel<false>), grid, block, sh, s, p);
  }
  B4R_CHECK_LAUNCH("b4r_attn_block_bwd");
  // gamma / beta gradients of the previous LayerNorm: ordered sum over the sequences (queued with the caller's reductions)
  return b4r_launch_slab_reduce_full(d->scratch, d->B, 1, 128, d->dprev_gamma, 128, 0, nullptr, nullptr, nullptr, nullptr, s);
}

int b4r_attn32_fwd(const b4r_attn_block_desc* d, b4r_stream_t stream);
bool b4r_attn32_core_preferred(int L);
extern "C" int b4r_attn_block_fwd(const b4r_attn_block_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_attn_block_fwd: null descriptor");
  static const bool fwd32 = !(getenv("B4R_ATTN32_FWD") && atoi(getenv("B4R_ATTN32_FWD")) == 0);
  // The 32-token-tile forward writes the attention-dropout decisions in the 32-key-tile layout ONLY.  Its readers are the
  // 32-token-tile block backward and, where a step's backward is unfused (L > 208, B4R_ATTN_BWD_FUSED=0, ...), the attention core's
  // backward -- which reads that layout only while b4r_attn32_core_preferred holds (B4R_ATTN32_CORE).  So the kernel is taken only
  // where BOTH readers read what it writes; otherwise the 16-token-tile forward below runs, which writes both layouts.  (The minimum
  // length of b4r_attn32_set_min_len enters both predicates: change it between steps, never between a forward and its backward.)
  if (fwd32 && use_attn32() && b4r_attn32_preferred(d->H, d->heads, d->L) && b4r_attn32_core_preferred(d->L)) return b4r_attn32_fwd(d, stream);
  B4R_CHECK_ARG(b4r_attn_block_supported(d->H, d->heads, d->L), B4R_E_SHAPE,
                "b4r_attn_block_fwd: needs hidden size 64, 2 heads, L <= 256 and the bf16x3 mode (H=%d heads=%d L=%d)", d->H, d->heads,
                d->L);
  const bool embed = d->emb_ids != nullptr;
  B4R_CHECK_ARG(d->B > 0 && (d->x || embed) && d->input_mask && d->Wqkv && d->bqkv && d->Wo && d->bo && d->ln_gamma && d->ln_beta &&
                    (d->x1 || (d->z1 && d->mean1 && d->rstd1)),
                B4R_E_BADARG, "b4r_attn_block_fwd: null argument (outputs: x1, or z1 + mean1 + rstd1)");
  B4R_CHECK_ARG(!embed || (d->emb_table && d->emb_pos && d->emb_gamma && d->emb_beta && d->emb_x && d->emb_vocab > 0 &&
                           al16(d->emb_table) && al16(d->emb_pos) && al16(d->emb_gamma) && al16(d->emb_beta) && al16(d->emb_x)),
                B4R_E_BADARG, "b4r_attn_block_fwd: the embedding mode needs emb_table, emb_pos, emb_gamma, emb_beta, emb_x (16-byte "
                "aligned) and emb_vocab");
  B4R_CHECK_ARG(al16(d->x) && al16(d->Wqkv) && al16(d->Wo) && al16(d->bo) && al16(d->ln_gamma) && al16(d->ln_beta) && al16(d->qkv) &&
                    al16(d->ctx) && al16(d->z1) && al16(d->x1) && al16(d->keep_bits),
                B4R_E_ALIGN, "b4r_attn_block_fwd: operands must be 16-byte aligned");
  AbP p{};
  p.x = d->x; p.mask = d->input_mask; p.Wqkv = d->Wqkv; p.bqkv = d->bqkv; p.Wo = d->Wo; p.bo = d->bo;
  p.g1 = d->ln_gamma; p.be1 = d->ln_beta; p.qkv = d->qkv; p.ctx = d->ctx; p.lse = d->lse; p.bits = d->keep_bits;
  p.z1 = d->z1; p.x1 = d->x1; p.mean1 = d->mean1; p.rstd1 = d->rstd1;
  p.B = d->B; p.L = d->L; p.KT = b4r_cdiv(d->L, 16);
  p.qscale = 1.0f / sqrtf(32.0f); p.eps = d->ln_eps;
  p.drop_p = b4r_make_drop(d->rng, d->probs_stream, d->probs_rate, d->rng != nullptr);
  p.drop_o = b4r_make_drop(d->rng, d->out_stream, d->out_rate, d->rng != nullptr);
  B4R_CHECK_ARG(!p.drop_p.rng || d->keep_bits, B4R_E_BADARG, "b4r_attn_block_fwd: attention dropout needs keep_bits");
  if (p.bits && use_attn32() && b4r_attn32_supported(d->H, d->heads, d->L)) p.bits32 = p.bits + b4r_attn_rx_keep_words(d->B, d->L, d->heads);
  if (embed) {
    p.ids = d->emb_ids; p.table = d->emb_table; p.pos = d->emb_pos; p.g0 = d->emb_gamma; p.be0 = d->emb_beta; p.V = d->emb_vocab;
    p.x_out = d->emb_x; p.mean0 = d->emb_mean; p.rstd0 = d->emb_rstd; p.eps0 = d->emb_eps;
    p.drop_e = b4r_make_drop(d->rng, d->emb_stream, d->emb_rate, d->rng != nullptr);
  }
  const int KTt = p.KT <= 4 ? 4 : p.KT <= 8 ? 8 : p.KT <= 13 ? 13 : 16;
  const dim3 grid((unsigned)d->B), block((unsigned)(64 * p.KT));
  hipStream_t s = (hipStream_t)stream;
  int rc;
#define AB_CASE(KT_)                                                                                   \
  case KT_:                                                                                            \
    rc = b4r_raise_lds((const void*)attn_block_fwd_kernel<KT_>, fwd_lds<KT_>(), "b4r_attn_block_fwd"); \
    if (rc) return rc;                                                                                 \
    hipLaunchKernelGGL((attn_block_fwd_kernel<KT_>), grid, block, fwd_lds<KT_>(), s, p);               \
    break;
  switch (KTt) {
    AB_CASE(4) AB_CASE(8) AB_CASE(13) AB_CASE(16)
    default: b4r_set_error("b4r_attn_block_fwd: internal"); return B4R_E_SHAPE;
  }
#undef AB_CASE
  B4R_CHECK_LAUNCH("b4r_attn_block_fwd");
  return B4R_OK;
}
