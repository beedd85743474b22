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
#include "b4r_rx_tiles.h"

namespace {

constexpr int SUB = 1024;     // bytes of one 16 x 32 bf16 sub-tile
constexpr int HID = 64;

__device__ __forceinline__ int sub_off(int r16, int ch) { return r16 * 64 + 16 * (ch ^ ((0 - (r16 >> 2)) & 3)); }
__device__ __forceinline__ int sub_base(int rt, int cb, int ncb) { return ((rt * ncb + cb) * 2) * SUB; }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
__device__ __forceinline__ bf16x8 tr_pair(const char* a, const char* b) {
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)b);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}

// W [R][C] fp32 row-major -> natural hi / lo image in 16 x 32 sub-tiles, all `nthreads` threads of the workgroup
__device__ __forceinline__ void stage_weight(char* img, const float* W, int R, int C, int nthreads) {
  const int c4n = C >> 2, ncb = C >> 5, nf4 = R * c4n;
  for (int f = threadIdx.x; f < nf4; f += nthreads) {
    const int r = f / c4n, c = 4 * (f - r * c4n);
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)r * C + c);
    bf16x4 h, l;
    b4r_split4(v, h, l);
    const int cc = c & 31;
    char* dst = img + sub_base(r >> 4, c >> 5, ncb) + sub_off(r & 15, cc >> 3) + 8 * ((cc >> 2) & 1);
    *reinterpret_cast<bf16x4*>(dst) = h;
    *reinterpret_cast<bf16x4*>(dst + SUB) = l;
  }
}

struct AbP {
  const float* x; const int64_t* mask;
  const float* Wqkv; const float* bqkv; const float* Wo; const float* bo; const float* g1; const float* be1;
  float* qkv; float* ctx; float* lse; uint32_t* bits;
  float* z1; float* x1; float* mean1; float* rstd1;
  int B, L, KT;
  float qscale, eps;
  DropArgs drop_p, drop_o;
};

__device__ __forceinline__ float sum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ float quad_sum(float s) {
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
__device__ __forceinline__ f32x4 lo4(const f32x8 v) { return (f32x4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ f32x4 hi4(const f32x8 v) { return (f32x4){v[4], v[5], v[6], v[7]}; }

// KTT: compile-time bound of the 16-key tiles (the score row lives in registers); block = 64 * KT threads, KT = ceil(L / 16)
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4, qq = i >> 2, pp = i & 3;

  // ---- phase 0: weights and the key mask ------------------------------------------------------------------------------
  const int tok = 16 * wave + i, tokc = min(tok, L - 1);
  f32x8 xv[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) xv[ks] = load8(p.x + (row0 + tokc) * HID + 32 * ks + 8 * g);   // in flight during the staging
  stage_weight(big, p.Wqkv, HID, 3 * HID, nthreads);
  stage_weight(woimg, p.Wo, HID, HID, nthreads);
  for (int k = threadIdx.x; k < KTE * 16; k += nthreads)
    sAdd[k] = k < L ? (1.0f - (float)p.mask[row0 + k]) * -1e9f : -INFINITY;
  for (int k = threadIdx.x; k < 3 * HID; k += nthreads) sbq[k] = p.bqkv[k];
  bf16x8 xh[2], xl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) split8(xv[ks], xh[ks], xl[ks]);
  __syncthreads();

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
  __syncthreads();   // every wave is done with the Wqkv image: the K / V images may overwrite it

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
  __syncthreads();

  // ---- phase 3: attention per head (attn_rx_fwd_kernel's body) --------------------------------------------------------------
  const FragAddr fa = frag_addr(lane);
  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  f32x4 o[2][2];
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    const char* img = big + hd * KTE * TILE_BYTES;
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
    if (g == 0 && live && p.lse) p.lse[bh * L + tok] = m + __logf(sum);
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
  const DropCtx dco = b4r_drop_ctx(p.drop_o);
  f32x4 z[4];
  float s = 0.f;
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const f32x4 v = y[hb] + *reinterpret_cast<const f32x4*>(p.bo + 16 * hb + 4 * g);
    const f32x4 res = *reinterpret_cast<const f32x4*>(p.x + (row0 + tokc) * HID + 16 * hb + 4 * g);
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
        out[e] = z[hb][e] * inv + (be[e] - mean * inv);
      }
      if (p.z1) *reinterpret_cast<f32x4*>(p.z1 + off) = z[hb];
      *reinterpret_cast<f32x4*>(p.x1 + off) = out;
    }
    if (g == 0) {
      if (p.mean1) p.mean1[row0 + tok] = mean;
      if (p.rstd1) p.rstd1[row0 + tok] = rstd;
    }
  }
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

extern "C" int32_t b4r_attn_block_supported(int32_t hidden_size, int32_t num_heads, int32_t L) {
  return (hidden_size == HID && num_heads == 2 && L > 0 && L <= 256 && b4r_get_gemm_mode() == B4R_GEMM_BF16X3) ? 1 : 0;
}

extern "C" int b4r_attn_block_fwd(const b4r_attn_block_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_attn_block_fwd: null descriptor");
  B4R_CHECK_ARG(b4r_attn_block_supported(d->H, d->heads, d->L), B4R_E_SHAPE,
                "b4r_attn_block_fwd: needs hidden size 64, 2 heads, L <= 256 and the bf16x3 mode (H=%d heads=%d L=%d)", d->H, d->heads,
                d->L);
  B4R_CHECK_ARG(d->B > 0 && d->x && d->input_mask && d->Wqkv && d->bqkv && d->Wo && d->bo && d->ln_gamma && d->ln_beta && d->x1,
                B4R_E_BADARG, "b4r_attn_block_fwd: null argument");
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
