// Device code shared by the two halves of the fused encoder layer (b4r_attn_block.hip, b4r_ffn_rx.hip): the natural-order bf16
// hi / lo LDS images in 16 x 32 sub-tiles, the lane constants of their fragment shapes and the feed-forward forward pass of one
// 16-token tile (used by ffn_fwd_kernel and, behind the attention half, by the whole-layer forward kernel).
// Reference: tfm TransformerEncoderBlock as built at bert4rec/models/components/networks/bert4rec_encoder.py:136-147, called :220-222.
#pragma once
#include "b4r_rx_tiles.h"

namespace {

constexpr int SUB = 1024;     // bytes of one 16 x 32 bf16 sub-tile
constexpr int HID = 64, INNER = 256;
constexpr int W_IMG = HID * INNER * 4;   // hi + lo image of one feed-forward weight matrix: 64 KB
// unroll factor of the forward's inner-dimension loop: what hipcc allocates without spilling into the loop at 128 VGPRs
#ifndef FFN_FWD_UNROLL
#define FFN_FWD_UNROLL 4
#endif
// timing experiments only (tools/build_variant.sh): 1 = no weight staging, 2 = no GELU arithmetic, 4 = no output stores
#ifndef FFN_EXP
#define FFN_EXP 0
#endif

__device__ __forceinline__ int sub_off(int r16, int ch) { return r16 * 64 + 16 * (ch ^ ((0 - (r16 >> 2)) & 3)); }
// hi sub-tile (row tile rt, column block cb) of an image with ncb column blocks; the lo sub-tile follows it
__device__ __forceinline__ int sub_base(int rt, int cb, int ncb) { return ((rt * ncb + cb) * 2) * SUB; }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
__device__ __forceinline__ bf16x8 tr_pair(const char* a, const char* b) {
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)a);
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)b);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 row_at(const char* a) { return *reinterpret_cast<const bf16x8*>(a); }

// W [R][C] fp32 row-major -> natural hi / lo image (R % 16 == 0, C % 32 == 0), all `nthreads` threads of the workgroup.
// One float4 per thread and turn, sub-tile by sub-tile: the 64 lanes of a wave-instruction fill 8 whole rows (512 bytes) of one
// sub-tile, so the 8-byte LDS stores are conflict-free (row-major order put a wave across 8 sub-tiles 2 KB apart: 8-way).
// Four turns are requested before the first is converted: one load latency per four turns instead of one per turn (the in-kernel
// stamps showed 900 cycles per turn, i.e. the loop waited for every load: 3.5 us for the two feed-forward matrices).
constexpr int STAGE_U = 4;
struct StageTurns { f32x4 v[STAGE_U]; int off[STAGE_U]; };
__device__ __forceinline__ void stage_request(StageTurns& t, const float* W, int R, int C, int f0, int nthreads) {
  const int ncb = C >> 5, nf4 = (R * C) >> 2;
#pragma unroll
  for (int u = 0; u < STAGE_U; ++u) {
    const int f = min(f0 + u * nthreads, nf4 - 1);   // clamped: every load unconditional
    const int st = f >> 7, w = f & 127;
    const int rt = st / ncb, cb = st - rt * ncb;
    const int r16 = w >> 3, q4 = w & 7;
    t.v[u] = *reinterpret_cast<const f32x4*>(W + (int64_t)(16 * rt + r16) * C + 32 * cb + 4 * q4);
    t.off[u] = sub_base(rt, cb, ncb) + sub_off(r16, q4 >> 1) + 8 * (q4 & 1);
  }
}
__device__ __forceinline__ void stage_commit(const StageTurns& t, char* img, int R, int C, int f0, int nthreads) {
  const int nf4 = (R * C) >> 2;
#pragma unroll
  for (int u = 0; u < STAGE_U; ++u) {
    if (f0 + u * nthreads < nf4) {
      bf16x4 h, l;
      b4r_split4(t.v[u], h, l);
      *reinterpret_cast<bf16x4*>(img + t.off[u]) = h;
      *reinterpret_cast<bf16x4*>(img + t.off[u] + SUB) = l;
    }
  }
}
__device__ __forceinline__ void stage_weight(char* img, const float* W, int R, int C, int nthreads) {
  const int nf4 = (R * C) >> 2;
  for (int f0 = threadIdx.x; f0 < nf4; f0 += STAGE_U * nthreads) {
    StageTurns t;
    stage_request(t, W, R, C, f0, nthreads);
    stage_commit(t, img, R, C, f0, nthreads);
  }
}
// two matrices: one after the other (requesting the turns of both together was measured: 3 us per step SLOWER)
__device__ __forceinline__ void stage_weight_pair(char* img_a, const float* Wa, int Ra, int Ca, char* img_b, const float* Wb, int Rb, int Cb,
                                                  int nthreads) {
  stage_weight(img_a, Wa, Ra, Ca, nthreads);
  stage_weight(img_b, Wb, Rb, Cb, nthreads);
}

__device__ __forceinline__ float sum4(const f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ float quad_sum(float s) {   // over the four lanes i, i+16, i+32, i+48 that share a token
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
// sum over the 16 lanes of a DPP row (the 16 tokens of a lane group), valid in lane 15 of the row: four v_add_f32 with a row_shr
// modifier, no LDS crossbar traffic (a __shfl_xor butterfly compiles to ds_bpermute_b32: 5 us in the attention backward's epilogue)
__device__ __forceinline__ float row_sum15(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
  return v;
}

struct LaneK {
  int tr[2];      // transposed fragment, rows = two stacked 16-row tiles (dw kernel): rows 4g .. 4g+3, column 16 db + i
  int trk[2][2];  // transposed fragment whose rows are a 32-deep k block in natural order: [db][s] = rows 8g + 4s .. +3 of the
                  // block (row tile g >> 1 of the pair: + (g >> 1) * row-tile stride, added by the caller), column 16 db + i
  int trw[2][2];  // the same rows, columns interleaved: [a][s] = rows 8g + 4s .., column 8p + 4a + e for lane i = 4p + e
  int row;        // row fragment: row i, columns 8g .. 8g+7
  int hi_tile;    // g >> 1: which row tile of a 32-row pair the lane's k rows are in
};
__device__ __forceinline__ LaneK lane_consts(int lane) {
  const int i = lane & 15, g = lane >> 4, qq = i >> 2, pp = i & 3;
  LaneK k;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    k.tr[db] = sub_off(4 * g + qq, 2 * db + (pp >> 1)) + 8 * (pp & 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      k.trk[db][s] = sub_off(8 * (g & 1) + 4 * s + qq, 2 * db + (pp >> 1)) + 8 * (pp & 1);
      k.trw[db][s] = sub_off(8 * (g & 1) + 4 * s + qq, pp) + 8 * db;
    }
  }
  k.row = sub_off(i, g);
  k.hi_tile = g >> 1;
  return k;
}

// fpre^T tile `a` of inner block kt (rows 4p + e = inner columns 32 kt + 8p + 4a + e, the wave's 16 tokens on the columns),
// accumulator preset to the bias `c` (sb1[32 kt + 8g + 4a ..], read by the caller an iteration ahead: a wait on it would
// otherwise drain every LDS read in flight in front of each product); xh / xl: the tokens' x1 rows as B operands, k-slot (g, j) = hidden column 32 ks + 8g + j
__device__ __forceinline__ f32x4 fpre_tile(const char* w1img, f32x4 c, const LaneK& lk, int kt, int a,
                                           const bf16x8 (&xh)[2], const bf16x8 (&xl)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const char* t = w1img + sub_base(2 * ks + lk.hi_tile, kt, 8);
    c = mfma3(tr_pair(t + lk.trw[a][0], t + lk.trw[a][1]), tr_pair(t + SUB + lk.trw[a][0], t + SUB + lk.trw[a][1]), xh[ks], xl[ks], c);
  }
  return c;
}

// the wave's 16 rows of a [N, 64] matrix as B operands of products that sum over the hidden index
__device__ __forceinline__ void load_rows(const float* src, int tokc, int g, f32x8 (&v)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) v[ks] = load8(src + (int64_t)tokc * HID + 32 * ks + 8 * g);
}

// The feed-forward half's input x1 = LayerNorm(z1) * g1 + be1 is either given ([N, 64]) or formed on load from the attention half's
// pre-LayerNorm sum z1 and its statistics (x1 == NULL: then the attention block need not store x1 at all -- 13 MB less written per
// layer at ML-1M -- and the kernels that read both z1 and x1 read one tensor).  The formula is the attention epilogue's.
struct X1Src {
  const float* x1; const float* z1; const float* mean1; const float* rstd1; const float* g1; const float* be1;
};
// 8 (or 4) consecutive columns col.. of row `tok`
__device__ __forceinline__ f32x8 x1_load8(const X1Src& s, int64_t tok, int col) {
  if (s.x1 != nullptr) return load8(s.x1 + tok * HID + col);
  const f32x8 z = load8(s.z1 + tok * HID + col), gm = load8(s.g1 + col), be = load8(s.be1 + col);
  const float mean = s.mean1[tok], rstd = s.rstd1[tok];
  f32x8 y;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float inv = rstd * gm[e];
    y[e] = fmaf(z[e], inv, fmaf(-mean, inv, be[e]));   // explicit: every kernel that forms x1 rounds alike
  }
  return y;
}
__device__ __forceinline__ f32x4 x1_load4(const X1Src& s, int64_t tok, int col) {
  if (s.x1 != nullptr) return *reinterpret_cast<const f32x4*>(s.x1 + tok * HID + col);
  const f32x4 z = *reinterpret_cast<const f32x4*>(s.z1 + tok * HID + col), gm = *reinterpret_cast<const f32x4*>(s.g1 + col),
              be = *reinterpret_cast<const f32x4*>(s.be1 + col);
  const float mean = s.mean1[tok], rstd = s.rstd1[tok];
  f32x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float inv = rstd * gm[e];
    y[e] = fmaf(z[e], inv, fmaf(-mean, inv, be[e]));   // explicit: every kernel that forms x1 rounds alike
  }
  return y;
}
// the wave's 16 rows of x1 as B operands of products that sum over the hidden index
__device__ __forceinline__ void load_x1_rows(const X1Src& s, int tok, int g, f32x8 (&v)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) v[ks] = x1_load8(s, tok, 32 * ks + 8 * g);
}

// what the forward pass of one tile reads and writes besides the weight images
struct FfnTileP {
  X1Src x1; const float* b2; const float* g2; const float* be2;
  float* z2; float* x2; float* mean2; float* rstd2;
  float eps;
};

// Feed-forward forward of the wave's 16 rows: x2 = LN(x1 + dropout(gelu(x1.W1 + b1).W2 + b2)).  Lane (i, g): row `tok` of the
// [N, 64] tensors (pad lanes carry a valid row and store == false); w1img / w2img / sb1: the staged W1, W2 images and b1 in LDS.
__device__ __forceinline__ void ffn_fwd_tile(const FfnTileP& p, const char* w1img, const char* w2img, const float* sb1, const LaneK& lk,
                                             const DropCtx& dctx, int tok, bool store, int g) {
  bf16x8 xh[2], xl[2];
  {
    f32x8 xv[2];
    load_x1_rows(p.x1, tok, g, xv);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) split8(xv[ks], xh[ks], xl[ks]);
  }
  f32x4 acc[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) acc[hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x8 bnext = load8(&sb1[8 * g]);
#pragma unroll FFN_FWD_UNROLL
  for (int kt = 0; kt < INNER / 32; ++kt) {
    const f32x8 bias = bnext;
    bnext = load8(&sb1[32 * min(kt + 1, INNER / 32 - 1) + 8 * g]);
    f32x4 f[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const f32x4 c = fpre_tile(w1img, a ? (f32x4){bias[4], bias[5], bias[6], bias[7]} : (f32x4){bias[0], bias[1], bias[2], bias[3]},
                                lk, kt, a, xh, xl);
#pragma unroll
      for (int r = 0; r < 4; ++r) f[a][r] = (FFN_EXP & 2) ? c[r] : b4r_gelu_fast(c[r]);
    }
    bf16x8 fh, fl;
    split8(cat(f[0], f[1]), fh, fl);   // k-slot (g, j) = inner column 32 kt + 8g + j
    const char* t = w2img + sub_base(2 * kt + lk.hi_tile, 0, 2);
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {   // G^T[16 hb + ..][token] += W2^T[.., inner block kt] . f^T
      const char* th = t + (hb >> 1) * 2 * SUB;
      const int db = hb & 1;
      acc[hb] = mfma3(tr_pair(th + lk.trk[db][0], th + lk.trk[db][1]), tr_pair(th + SUB + lk.trk[db][0], th + SUB + lk.trk[db][1]),
                      fh, fl, acc[hb]);
    }
  }
  // bias + dropout + residual + LayerNorm: lane (i, g) holds columns 16 hb + 4g .. +3 of token i
  f32x4 z[4];
  float s = 0.f;
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const f32x4 y = acc[hb] + *reinterpret_cast<const f32x4*>(p.b2 + 16 * hb + 4 * g);
    const f32x4 res = x1_load4(p.x1, tok, 16 * hb + 4 * g);
    z[hb] = res + b4r_drop4(dctx, y, (uint64_t)tok * HID + (uint64_t)(16 * hb + 4 * g));
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
  if (store && (!(FFN_EXP & 4) || rstd == 12345.f)) {
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      const int64_t o = (int64_t)tok * HID + 16 * hb + 4 * g;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(p.g2 + 16 * hb + 4 * g);
      const f32x4 be = *reinterpret_cast<const f32x4*>(p.be2 + 16 * hb + 4 * g);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float inv = rstd * gm[e];
        y[e] = z[hb][e] * inv + (be[e] - mean * inv);
      }
      if (p.z2) *reinterpret_cast<f32x4*>(p.z2 + o) = z[hb];
      *reinterpret_cast<f32x4*>(p.x2 + o) = y;
    }
    if (g == 0) {
      if (p.mean2) p.mean2[tok] = mean;
      if (p.rstd2) p.rstd2[tok] = rstd;
    }
  }
}

}  // namespace
