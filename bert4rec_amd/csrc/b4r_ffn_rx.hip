// Feed-forward half of a Keras TransformerEncoderBlock (post-LN) as THREE kernels that keep the [N, inner] intermediate on
// the chip (hidden size 64, inner size 256, split-precision bf16x3 arithmetic of b4r_rx_tiles.h):
//
//   forward   x2 = LN(x1 + dropout(gelu(x1.W1 + b1).W2 + b2))                     ffn_fwd_kernel
//   backward  dz1 = LN1'( (dropmask(dz2).W2^T * gelu'(x1.W1 + b1)).W1^T + dz2 )    ffn_bwd_dx_kernel   (recomputes the pre-activation)
//             dW2 = gelu(..)^T.dropmask(dz2), dW1 = x1^T.dFpre, db1, db2           ffn_bwd_dw_kernel   (recomputes it again)
//
// Reference: tfm TransformerEncoderBlock as constructed at bert4rec/models/components/networks/bert4rec_encoder.py:136-147 and
// called at :220-222 (intermediate dense + erf-GELU, output dense, output dropout, residual, output_layer_norm).
//
// Why three kernels and not the five launches of round 1 (FFN-in, FFN-out + LN | pair kernel, dX1 + LN', dW1): the
// [N, 256] pre-activation / activation / their gradients were written and re-read through HBM seven times per layer
// (~565 MB of the layer's ~1.3 GB); here they only ever exist as accumulator tiles.  The price is recomputing x1.W1 in
// both backward kernels (2 x 1.7 GFLOP at ML-1M), which the matrix pipe has to spare.
//
// Orientation.  v_mfma_f32_16x16x32_bf16 leaves D[4g + r][i] in register r of lane (i = lane & 15, g = lane >> 4).  A product
// whose result feeds the NEXT product as its B operand (k on (g, j), column on i) must therefore be computed so that the
// index the next product sums over lands on D's rows:
//   * forward / dx:  everything transposed, tokens on i.  fpre^T = W1^T.x1^T puts the inner index on D's rows, so
//     G^T = W2^T.f^T (forward) and dx1^T = W1.dfpre^T (backward) take the activation tiles straight from the accumulators,
//     two 16-row tiles per 32-deep instruction: k-slot (g, j) = row 4g + j of the first tile for j < 4, of the second for
//     j >= 4 (the attention kernels' convention).  A wave owns 16 tokens end to end: no barrier, no LDS traffic for
//     activations, the LayerNorm statistics are two 4-lane shuffles.
//   * dw:  tokens are the summation index, so fpre / dF are computed un-transposed (tokens on D's rows, the wave's 16
//     inner columns on i) and feed dW2^T = dG^T.f and dW1 = x1^T.dFpre as B operands; a wave owns 16 inner columns (its
//     slices of W1 / W2 live in registers), the 16 waves of a workgroup share 32-token chunks of x1 / dropmask(dz2) staged
//     as images.
// Weights live in LDS as natural-order bf16 hi / lo images cut in 16 x 32 sub-tiles (b4r_rx_tiles.h's swizzle); one image
// serves both fragment shapes: transposed (ds_read_b64_tr_b16: W^T as the A operand) and row (ds_read_b128: W as the A
// operand).  So that the row reads stay 16-byte reads, the two fpre^T tiles of a 32-wide inner block interleave its columns
// in groups of four (tile a holds inner columns 8p + 4a + e, p, e < 4, on its rows 4p + e): stacked as a B operand, k-slot
// (g, j) is then inner column 8g + j -- the natural order a row fragment delivers.  (An earlier layout with contiguous
// tiles needed two 8-byte reads per fragment; hipcc fused the hi / lo pair into ds_read2st64_b64, which banks modulo 32 and
// ran the input-gradient kernel at 72 % LDS bank-conflict cycles.)
#include "b4r_block_tiles.h"

namespace {

constexpr int FW = 16;        // waves per workgroup (1024 threads, one workgroup per CU: 129 KB of weight images)
// unroll factor of the input-gradient kernel's inner-dimension loop: what hipcc allocates without spilling at 128 VGPRs (16 waves)
#ifndef FFN_DX_UNROLL
#define FFN_DX_UNROLL 1
#endif
// gelu(x) and gelu'(x) from one erf / exp evaluation (b4r_erf_as)
__device__ __forceinline__ void gelu_both(float x, float& gl, float& gr) {
  float e;
  const float er = b4r_erf_as(x * 0.70710678118654752440f, e);
  const float cdf = 0.5f * (1.0f + er);
  gl = x * cdf;
  gr = fmaf(x * 0.39894228040143267794f, e, cdf);
}

// FFN_PROF (tools/build_variant.sh ... -DFFN_PROF): thread 0 of workgroup 0 stamps the shader clock at phase boundaries
#ifdef FFN_PROF
__device__ long long g_ff_prof[64];
#define FF_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_ff_prof[k] = clock64(); } while (0)
#else
#define FF_MARK(k) do { } while (0)
#endif
struct FfnP {
  const float* x1; const float* W1; const float* b1; const float* W2; const float* b2;
  const float* g2; const float* be2;
  const float* be1;   // with x1 == NULL the input is formed from z1, mean1, rstd1, g1, be1 (X1Src)
  float* z2; float* x2; float* mean2; float* rstd2;
  const float* dz2; const float* z1; const float* mean1; const float* rstd1; const float* g1;
  float* dz1; float* ln_part;
  float* slab_w1; float* slab_b1; float* slab_w2; float* slab_b2;
  int N; float eps;
  DropArgs drop;
  // row list (the last layer of a train step: only the rows the masked-LM head gathers carry a gradient, b4r_mlm_rows):
  // compact index j -> row rows[j], j < *n_dev.  slotof[j]: the valid masked-LM slot of row j (-1: none), dgr [M,64] the
  // gradient per slot; then dz2 is formed in the kernel as the output LayerNorm's backward of that row (z2 / mean2 / rstd2 / g2)
  const int* rows; const int* n_dev; const int* slotof; const float* dgr;
  // ... or the list in its implicit form (slot mode): entry j = masked-LM slot j of n_slots, row (j / sP) * sL + clamp(spos[j]),
  // slot j where sids[j] != 0
  const int64_t* spos; const int64_t* sids; int sP, sL, n_slots;
  float* dz2c;      // [cap,64] compact dz2 for the weight-gradient kernel
  float* ln2_part;  // [grid][128] gamma / beta partials of the output LayerNorm
};

// number of (compact) rows and the actual row of compact index j
__device__ __forceinline__ int ffn_rows(const FfnP& p) { return p.spos ? p.n_slots : (p.n_dev ? *p.n_dev : p.N); }
__device__ __forceinline__ int ffn_row(const FfnP& p, int j) {
  if (p.spos) {
    const int64_t q = p.spos[j];
    return (j / p.sP) * p.sL + (q < 0 ? 0 : (q >= p.sL ? p.sL - 1 : (int)q));   // clamped like b4r_gather_rows
  }
  return p.rows ? p.rows[j] : j;
}
__device__ __forceinline__ bool ffn_listed(const FfnP& p) { return p.rows != nullptr || p.spos != nullptr; }
__device__ __forceinline__ int ffn_slot(const FfnP& p, int j) { return p.spos ? (p.sids[j] != 0 ? j : -1) : p.slotof[j]; }

// -----------------------------------------------------------------------------------------------------------
// forward.  LDS: [W1 image 64 KB | W2 image 64 KB | b1]
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * FW) void ffn_fwd_kernel(FfnP p) {
  FF_MARK(0);
  extern __shared__ __attribute__((aligned(16))) char smem_ffn[];
  char* w1img = smem_ffn;
  char* w2img = smem_ffn + W_IMG;
  float* sb1 = reinterpret_cast<float*>(smem_ffn + 2 * W_IMG);
  if (!(FFN_EXP & 1)) {
  stage_weight_pair(w1img, p.W1, HID, INNER, w2img, p.W2, INNER, HID, 64 * FW);
  }
  if (threadIdx.x < INNER) sb1[threadIdx.x] = p.b1[threadIdx.x];
  FF_MARK(1);
  __syncthreads();
  FF_MARK(2);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const LaneK lk = lane_consts(lane);
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  const int Nn = ffn_rows(p);
  const int ntiles = (Nn + 15) >> 4;
  const FfnTileP tp{X1Src{p.x1, p.z1, p.mean1, p.rstd1, p.g1, p.be1}, p.b2, p.g2, p.be2, p.z2, p.x2, p.mean2, p.rstd2, p.eps};
  for (int t = blockIdx.x + gridDim.x * wave; t < ntiles; t += gridDim.x * FW) {   // wave-uniform: EXEC stays full
    const int j = 16 * t + i;
    const int tok = ffn_row(p, min(j, Nn - 1));   // the row in the [N, 64] tensors; pad lanes repeat the last row
    FF_MARK(3);
    ffn_fwd_tile(tp, w1img, w2img, sb1, lk, dctx, tok, j < Nn, g);
    FF_MARK(6);
  }
  FF_MARK(7);
}

// -----------------------------------------------------------------------------------------------------------
// backward, input gradient + the attention LayerNorm's backward.  LDS: [W1 image | W2 image | b1 | LayerNorm partials]
// -----------------------------------------------------------------------------------------------------------
// NW waves per workgroup: 16 (a 128-register cap, 10 registers spilled; the default) or 8 (B4R_FFN_DX_WAVES=8: two waves per SIMD, 256
// registers, no spill, a wave walks two 16-token tiles at ML-1M -- 3-4 us slower per layer, see the launch site)
template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void ffn_bwd_dx_kernel(FfnP p) {
  FF_MARK(10);
  extern __shared__ __attribute__((aligned(16))) char smem_ffn[];
  char* w1img = smem_ffn;
  char* w2img = smem_ffn + W_IMG;
  float* sb1 = reinterpret_cast<float*>(smem_ffn + 2 * W_IMG);
  float* sred = sb1 + INNER;   // [FW][128], then [FW][128] for the output LayerNorm (row-list mode)
  stage_weight_pair(w1img, p.W1, HID, INNER, w2img, p.W2, INNER, HID, 64 * NW);
  if (threadIdx.x < INNER) sb1[threadIdx.x] = p.b1[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const LaneK lk = lane_consts(lane);
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  const int Nn = ffn_rows(p);
  const int ntiles = (Nn + 15) >> 4;
  const bool rowmode = p.slotof != nullptr || p.spos != nullptr;
  // LayerNorm gamma / beta sums of this wave live in its 128 floats of LDS (32 more live registers per lane spilled 37 VGPRs to
  // scratch: 30 MB of extra HBM writes per launch in profiles/r02_a)
  float* myred = sred + wave * 128;
  float* myred2 = sred + (NW + wave) * 128;
  myred[lane] = 0.f;
  myred[64 + lane] = 0.f;
  myred2[lane] = 0.f;
  myred2[64 + lane] = 0.f;
  FF_MARK(11);

  for (int t = blockIdx.x + gridDim.x * wave; t < ntiles; t += gridDim.x * NW) {
    const int j = 16 * t + i, jc = min(j, Nn - 1);
    const int tok = ffn_row(p, jc), tokc = tok;
    const int64_t rowo = (int64_t)tokc * HID + 4 * g;
    // row-list mode: dz2 = LN2'(dx2) with dx2 = the head's gradient of this row's slot (zero if the row only serves padded slots)
    const int slot = rowmode ? ffn_slot(p, jc) : -1;
    float ln2_mean = 0.f, ln2_rstd = 0.f, ln2_c1 = 0.f, ln2_c2 = 0.f;
    bf16x8 xh[2], xl[2], gh[2], gl[2];
    {
      f32x8 xv[2], dg[2];
      load_x1_rows(X1Src{p.x1, p.z1, p.mean1, p.rstd1, p.g1, p.be1}, tokc, g, xv);
      if (rowmode) {
        ln2_mean = p.mean2[tokc]; ln2_rstd = p.rstd2[tokc];
        f32x8 zz[2], gm[2];
        load_rows(p.z2, tokc, g, zz);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          dg[ks] = slot >= 0 ? load8(p.dgr + (int64_t)slot * HID + 32 * ks + 8 * g) : zero8;
          gm[ks] = load8(p.g2 + 32 * ks + 8 * g);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xe = (zz[ks][e] - ln2_mean) * ln2_rstd, ge = dg[ks][e] * gm[ks][e];
            zz[ks][e] = xe; dg[ks][e] = ge;
            s1 += ge; s2 += ge * xe;
          }
        }
        ln2_c1 = quad_sum(s1) * (1.0f / HID); ln2_c2 = quad_sum(s2) * (1.0f / HID);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) dg[ks][e] = ln2_rstd * (dg[ks][e] - ln2_c1 - zz[ks][e] * ln2_c2);
      } else {
        load_rows(p.dz2, tokc, g, dg);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (dctx.on) {
          const uint64_t e0 = (uint64_t)tok * HID + (uint64_t)(32 * ks + 8 * g);
          const f32x4 lo4 = b4r_drop4(dctx, (f32x4){dg[ks][0], dg[ks][1], dg[ks][2], dg[ks][3]}, e0);
          const f32x4 hi4 = b4r_drop4(dctx, (f32x4){dg[ks][4], dg[ks][5], dg[ks][6], dg[ks][7]}, e0 + 4);
          dg[ks] = cat(lo4, hi4);
        }
        split8(xv[ks], xh[ks], xl[ks]);
        split8(dg[ks], gh[ks], gl[ks]);
      }
    }
    FF_MARK(12);
    f32x4 acc[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) acc[hb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x8 bnext = load8(&sb1[8 * g]);
#pragma unroll FFN_DX_UNROLL
    for (int kt = 0; kt < INNER / 32; ++kt) {
      const f32x8 bias = bnext;
      bnext = load8(&sb1[32 * min(kt + 1, INNER / 32 - 1) + 8 * g]);
      f32x4 dfp[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const f32x4 c = fpre_tile(w1img, a ? (f32x4){bias[4], bias[5], bias[6], bias[7]} : (f32x4){bias[0], bias[1], bias[2], bias[3]},
                                  lk, kt, a, xh, xl);
        // dF^T[inner][token] = W2[inner][:] . dG^T for the same inner columns as the fpre tile: row i = 4p + e of the
        // operand is row 8p + 4a + e of the 32-row block of W2
        const int wrow = 8 * (i >> 2) + 4 * a + (i & 3);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const char* s0 = w2img + sub_base(2 * kt + (wrow >> 4), ks, 2) + sub_off(wrow & 15, g);
          d = mfma3(row_at(s0), row_at(s0 + SUB), gh[ks], gl[ks], d);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dfp[a][r] = d[r] * b4r_gelu_grad_fast(c[r]);
      }
      bf16x8 ph, pl;
      split8(cat(dfp[0], dfp[1]), ph, pl);   // k-slot (g, j) = inner column 32 kt + 8g + j
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {   // dx1^T[16 hb + ..][token] += W1[.., inner block kt] . dFpre^T
        const char* s0 = w1img + sub_base(hb, kt, 8) + lk.row;
        acc[hb] = mfma3(row_at(s0), row_at(s0 + SUB), ph, pl, acc[hb]);
      }
    }
    FF_MARK(13);
    // dx1 = acc + dz2 (the residual branch), then back through x1 = LN(z1)
    const float mean = p.mean1[tokc], rstd = p.rstd1[tokc];
    const bool live = j < Nn;
    f32x4 ge[4], xhat[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      f32x4 dz2e;   // dz2 of this lane's output columns 16 hb + 4g .. (the residual branch)
      if (rowmode) {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 dx2 = slot >= 0 ? *reinterpret_cast<const f32x4*>(p.dgr + (int64_t)slot * HID + 16 * hb + 4 * g) : zero4;
        const f32x4 xh2 = (*reinterpret_cast<const f32x4*>(p.z2 + rowo + 16 * hb) - ln2_mean) * ln2_rstd;
        const f32x4 g2v = *reinterpret_cast<const f32x4*>(p.g2 + 16 * hb + 4 * g);
        dz2e = (dx2 * g2v - ln2_c1 - xh2 * ln2_c2) * ln2_rstd;
        // gamma / beta sums of the output LayerNorm, and the compact dz2 for the weight-gradient kernel
        f32x4 pg2 = live ? dx2 * xh2 : zero4, pb2 = live ? dx2 : zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pg2[e] = row_sum15(pg2[e]); pb2[e] = row_sum15(pb2[e]); }
        if (i == 15) {
          f32x4* rg = reinterpret_cast<f32x4*>(myred2 + 16 * hb + 4 * g);
          f32x4* rb = reinterpret_cast<f32x4*>(myred2 + 64 + 16 * hb + 4 * g);
          *rg = *rg + pg2;
          *rb = *rb + pb2;
        }
        if (live) *reinterpret_cast<f32x4*>(p.dz2c + (int64_t)j * HID + 16 * hb + 4 * g) = dz2e;
      } else {
        dz2e = *reinterpret_cast<const f32x4*>(p.dz2 + rowo + 16 * hb);
      }
      const f32x4 dx = acc[hb] + dz2e;
      const f32x4 zz = *reinterpret_cast<const f32x4*>(p.z1 + rowo + 16 * hb);
      const f32x4 gm = *reinterpret_cast<const f32x4*>(p.g1 + 16 * hb + 4 * g);
      xhat[hb] = (zz - mean) * rstd;
      ge[hb] = dx * gm;
      s1 += sum4(ge[hb]);
      s2 += sum4(ge[hb] * xhat[hb]);
      // column sums over the tile's 16 tokens (DPP row sums, lane 15 of every lane group adds them to the wave's LDS strip)
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 pg = live ? dx * xhat[hb] : zero4, pb = live ? dx : zero4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pg[e] = row_sum15(pg[e]); pb[e] = row_sum15(pb[e]); }
      if (i == 15) {
        f32x4* rg = reinterpret_cast<f32x4*>(myred + 16 * hb + 4 * g);
        f32x4* rb = reinterpret_cast<f32x4*>(myred + 64 + 16 * hb + 4 * g);
        *rg = *rg + pg;
        *rb = *rb + pb;
      }
    }
    const float c1 = quad_sum(s1) * (1.0f / HID), c2 = quad_sum(s2) * (1.0f / HID);
    FF_MARK(14);
    if (live && (!rowmode || slot >= 0)) {   // an entry without a slot has dz1 = 0 exactly and may repeat another entry's row
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        f32x4 dz;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = rstd * (ge[hb][e] - c1 - xhat[hb][e] * c2);
        *reinterpret_cast<f32x4*>(p.dz1 + (int64_t)tok * HID + 16 * hb + 4 * g) = dz;
      }
    }
  }
  FF_MARK(15);
  // LayerNorm gamma / beta partial sums of the workgroup: the waves' strips in order
  __syncthreads();
  FF_MARK(16);
  if (threadIdx.x < 128) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += sred[w * 128 + threadIdx.x];
    p.ln_part[(int64_t)blockIdx.x * 128 + threadIdx.x] = r;
  } else if (threadIdx.x < 256 && rowmode) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += sred[(NW + w) * 128 + threadIdx.x - 128];
    p.ln2_part[(int64_t)blockIdx.x * 128 + threadIdx.x - 128] = r;
  }
}

// -----------------------------------------------------------------------------------------------------------
// backward, weight gradients.  Wave w owns inner columns 16 w .. 16 w + 15; the workgroup walks 32-token chunks.
// LDS: 2 stages x [x1 image 8 KB | dropmask(dz2) image 8 KB]
// -----------------------------------------------------------------------------------------------------------
constexpr int CH_TOK = 32;
constexpr int CH_IMG = CH_TOK * HID * 4;   // hi + lo image of one [32, 64] chunk: 8 KB

__global__ __launch_bounds__(64 * FW) void ffn_bwd_dw_kernel(FfnP p) {
  __shared__ __attribute__((aligned(16))) char smem_dw[4 * CH_IMG];
  FF_MARK(30);
  const int lane = threadIdx.x & 63, ib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const LaneK lk = lane_consts(lane);
  const DropCtx dctx = b4r_drop_ctx(p.drop);

  // the wave's weight slices as B operands (k = hidden index in natural order 32 ks + 8g + j, column = inner 16 ib + i)
  bf16x8 w1h[2], w1l[2], w2h[2], w2l[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    f32x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = p.W1[(int64_t)(32 * ks + 8 * g + j) * INNER + 16 * ib + i];
    split8(a, w1h[ks], w1l[ks]);
    split8(load8(p.W2 + (int64_t)(16 * ib + i) * HID + 32 * ks + 8 * g), w2h[ks], w2l[ks]);
  }
  const float b1v = p.b1[16 * ib + i];

  // staging role of this thread: threads 0..511 carry x1, 512..1023 carry dz2; one float4 of the chunk each
  const int sid = threadIdx.x & 511, stok = sid >> 4, sc4 = sid & 15;
  const bool is_dz = threadIdx.x >= 512;
  const bool rowmode = ffn_listed(p);
  const float* ssrc = is_dz ? (rowmode ? p.dz2c : p.dz2) : p.x1;
  const int Nn = ffn_rows(p);
  const int soff = (is_dz ? CH_IMG : 0) + sub_base(stok >> 4, sc4 >> 3, 2) + sub_off(stok & 15, (sc4 & 7) >> 1) + 8 * (sc4 & 1);
  const int nchunks = (Nn + CH_TOK - 1) / CH_TOK;
  f32x4 sv;
  int srow = 0;   // the row in the [N, 64] tensors (dropout index) of the staged piece
  f32x4 db2 = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int c) {
    const int j = min(CH_TOK * c + stok, Nn - 1);
    srow = ffn_row(p, j);
    const int64_t src_row = (is_dz && rowmode) ? j : srow;   // the compact dz2 is indexed by j
    sv = (is_dz || p.x1 != nullptr) ? *reinterpret_cast<const f32x4*>(ssrc + src_row * HID + 4 * sc4)
                                    : x1_load4(X1Src{nullptr, p.z1, p.mean1, p.rstd1, p.g1, p.be1}, srow, 4 * sc4);
  };
  auto put = [&](int c, int stage) {
    const int j = CH_TOK * c + stok;
    f32x4 v = sv;
    if (is_dz) v = b4r_drop4(dctx, v, (uint64_t)srow * HID + (uint64_t)(4 * sc4));
    if (j >= Nn) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (is_dz) db2 += v;
    bf16x4 h, l;
    b4r_split4(v, h, l);
    char* dst = smem_dw + stage * 2 * CH_IMG + soff;
    *reinterpret_cast<bf16x4*>(dst) = h;
    *reinterpret_cast<bf16x4*>(dst + SUB) = l;
  };

  f32x4 dw1[4], dw2[4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) { dw1[hb] = (f32x4){0.f, 0.f, 0.f, 0.f}; dw2[hb] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float db1 = 0.f;

  FF_MARK(31);
  int c = blockIdx.x;
  if (c < nchunks) { fetch(c); put(c, 0); }
  __syncthreads();
  FF_MARK(32);
  for (int it = 0; c < nchunks; c += gridDim.x, ++it) {
    const int cn = c + gridDim.x;
    if (cn < nchunks) fetch(cn);   // in flight while this chunk is multiplied
    const char* ximg = smem_dw + (it & 1) * 2 * CH_IMG;
    const char* gimg = ximg + CH_IMG;
    f32x4 f[2], dfp[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      f32x4 pre = {b1v, b1v, b1v, b1v}, d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char* xa = ximg + sub_base(tt, ks, 2) + lk.row;
        const char* ga = gimg + sub_base(tt, ks, 2) + lk.row;
        pre = mfma3(row_at(xa), row_at(xa + SUB), w1h[ks], w1l[ks], pre);   // fpre[token][inner] = x1.W1 + b1
        d = mfma3(row_at(ga), row_at(ga + SUB), w2h[ks], w2l[ks], d);       // dF[token][inner] = dG.W2^T
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float gl_, gr_;
        gelu_both(pre[r], gl_, gr_);
        f[tt][r] = gl_;
        dfp[tt][r] = d[r] * gr_;
      }
      db1 += sum4(dfp[tt]);
    }
    bf16x8 fh, fl, ph, pl;
    split8(cat(f[0], f[1]), fh, fl);
    split8(cat(dfp[0], dfp[1]), ph, pl);
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
      const char* g0 = gimg + sub_base(0, hb >> 1, 2) + lk.tr[hb & 1];
      const char* x0 = ximg + sub_base(0, hb >> 1, 2) + lk.tr[hb & 1];
      const int nt = 2 * 2 * SUB;   // the chunk's second 16-token tile
      dw2[hb] = mfma3(tr_pair(g0, g0 + nt), tr_pair(g0 + SUB, g0 + SUB + nt), fh, fl, dw2[hb]);   // dW2^T[h][inner] += dG^T.f
      dw1[hb] = mfma3(tr_pair(x0, x0 + nt), tr_pair(x0 + SUB, x0 + SUB + nt), ph, pl, dw1[hb]);   // dW1[h][inner] += x1^T.dFpre
    }
    if (cn < nchunks) put(cn, (it + 1) & 1);
    __syncthreads();
    if (it < 8) FF_MARK(33 + it);
  }
  FF_MARK(41);

  // db2[h] = sum over tokens of dropmask(dz2): per-thread sums over chunks, then over the 32 token slots of the staging layout.
  // This reduction (one barrier) comes BEFORE the 131 KB of slab stores: behind them the barrier would wait for their acknowledgements.
  const int64_t wg = blockIdx.x;
  float* red = reinterpret_cast<float*>(smem_dw);   // [32][64]; the loop's last barrier has passed
  if (is_dz) *reinterpret_cast<f32x4*>(&red[stok * HID + 4 * sc4]) = db2;
  __syncthreads();
  if (threadIdx.x < HID) {
    float r = 0.f;
#pragma unroll
    for (int s = 0; s < CH_TOK; ++s) r += red[s * HID + threadIdx.x];
    p.slab_b2[wg * HID + threadIdx.x] = r;
  }
  // partial results of this workgroup -> slabs (summed over the workgroups in slab order by the deferred reduction)
  float* sw1 = p.slab_w1 + wg * (HID * INNER);
  float* sw2 = p.slab_w2 + wg * (HID * INNER);
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    *reinterpret_cast<f32x4*>(sw2 + (int64_t)(16 * ib + i) * HID + 16 * hb + 4 * g) = dw2[hb];   // dW2 [inner][h]
#pragma unroll
    for (int r = 0; r < 4; ++r) sw1[(int64_t)(16 * hb + 4 * g + r) * INNER + 16 * ib + i] = dw1[hb][r];   // dW1 [h][inner]
  }
  db1 = quad_sum(db1);
  if (g == 0) p.slab_b1[wg * INNER + 16 * ib + i] = db1;
  FF_MARK(42);
}

int ffn_grid(int units) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0 || cus > 256) cus = 256;   // the scratch layout assumes at most 256 partial slabs
  }
  return units < cus ? units : cus;
}

constexpr size_t FWD_LDS = 2 * W_IMG + INNER * sizeof(float);
constexpr size_t DX_LDS = FWD_LDS + 2 * FW * 128 * sizeof(float);

FfnP make_p(const b4r_ffn_desc* d) {
  FfnP p{};
  p.x1 = d->x1; p.W1 = d->W1; p.b1 = d->b1; p.W2 = d->W2; p.b2 = d->b2; p.g2 = d->ln_gamma; p.be2 = d->ln_beta;
  p.be1 = d->ln1_beta;
  p.z2 = d->z2; p.x2 = d->x2; p.mean2 = d->mean2; p.rstd2 = d->rstd2;
  p.dz2 = d->dz2; p.z1 = d->z1; p.mean1 = d->mean1; p.rstd1 = d->rstd1; p.g1 = d->ln1_gamma; p.dz1 = d->dz1;
  p.N = d->N; p.eps = d->ln_eps;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, d->rng != nullptr);
  p.rows = d->rows; p.n_dev = d->n_rows;
  p.spos = d->slot_positions; p.sids = d->slot_ids; p.sP = d->slots_per_seq; p.sL = d->seq_len; p.n_slots = d->max_rows;
  return p;
}

bool al16(const void* q) { return q == nullptr || b4r_aligned16(q); }
// the block input: x1 itself, or everything needed to form it from the attention half's pre-LayerNorm sum
bool x1_given(const b4r_ffn_desc* d) {
  return d->x1 != nullptr || (d->z1 && d->mean1 && d->rstd1 && d->ln1_gamma && d->ln1_beta && b4r_aligned16(d->z1) &&
                              b4r_aligned16(d->ln1_gamma) && b4r_aligned16(d->ln1_beta));
}

// slot mode: all of its fields or none, never together with an explicit list, the rows it names inside [0, N)
bool slot_mode_ok(const b4r_ffn_desc* d) {
  if (d->slot_positions == nullptr) return d->slot_ids == nullptr;
  return d->slot_ids && d->rows == nullptr && d->n_rows == nullptr && d->slots_per_seq > 0 && d->seq_len > 0 && d->max_rows > 0 &&
         d->max_rows % d->slots_per_seq == 0 && (int64_t)(d->max_rows / d->slots_per_seq) * d->seq_len <= d->N;
}

}  // namespace

#ifdef FFN_PROF
extern "C" int b4r_debug_ff_prof(long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ff_prof), 64 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
#endif
extern "C" int32_t b4r_ffn_block_supported(int32_t hidden_size, int32_t inner_dim) {
  return (hidden_size == HID && inner_dim == INNER && b4r_get_gemm_mode() == B4R_GEMM_BF16X3) ? 1 : 0;
}

// per workgroup: dW1 + dW2 slabs, db1 + db2 strips, 128 LayerNorm partials
extern "C" int64_t b4r_ffn_block_bwd_scratch_floats(int32_t N) {
  const int64_t slabs = 256;   // an upper bound of the grid (one workgroup per CU)
  (void)N;
  return slabs * (2 * HID * INNER + INNER + HID + 128 + 128);
}

extern "C" int b4r_ffn_block_fwd(const b4r_ffn_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_ffn_block_fwd: null descriptor");
  B4R_CHECK_ARG(b4r_ffn_block_supported(d->H, d->I), B4R_E_SHAPE,
                "b4r_ffn_block_fwd: needs hidden size 64, inner size 256 and the bf16x3 mode (H=%d I=%d)", d->H, d->I);
  B4R_CHECK_ARG(d->N > 0 && x1_given(d) && d->W1 && d->b1 && d->W2 && d->b2 && d->ln_gamma && d->ln_beta && d->x2, B4R_E_BADARG,
                "b4r_ffn_block_fwd: null argument (the input is x1, or z1 + mean1 + rstd1 + ln1_gamma + ln1_beta)");
  B4R_CHECK_ARG(al16(d->x1) && al16(d->W1) && al16(d->W2) && al16(d->b2) && al16(d->ln_gamma) && al16(d->ln_beta) && al16(d->z2) &&
                    al16(d->x2),
                B4R_E_ALIGN, "b4r_ffn_block_fwd: operands must be 16-byte aligned");
  B4R_CHECK_ARG((d->rows == nullptr) == (d->n_rows == nullptr) && (d->rows == nullptr || d->max_rows > 0), B4R_E_BADARG,
                "b4r_ffn_block_fwd: rows, n_rows and max_rows go together");
  B4R_CHECK_ARG(slot_mode_ok(d), B4R_E_BADARG,
                "b4r_ffn_block_fwd: the slot mode needs slot_positions, slot_ids, slots_per_seq, seq_len, max_rows and no explicit list");
  const FfnP p = make_p(d);
  const int units = (d->rows || d->slot_positions) ? d->max_rows : d->N;
  int rc = b4r_raise_lds((const void*)ffn_fwd_kernel, FWD_LDS, "b4r_ffn_block_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(ffn_fwd_kernel, dim3(ffn_grid(b4r_cdiv(units, 16))), dim3(64 * FW), FWD_LDS, (hipStream_t)stream, p);
  B4R_CHECK_LAUNCH("b4r_ffn_block_fwd");
  return B4R_OK;
}

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);

// after_dx (optional): recorded between the two kernels -- dz1 is complete there, the weight gradients are not
int b4r_ffn_block_bwd_marked(const b4r_ffn_desc* d, hipStream_t stream, hipEvent_t after_dx);
extern "C" int b4r_ffn_block_bwd(const b4r_ffn_desc* d, b4r_stream_t stream) {
  return b4r_ffn_block_bwd_marked(d, (hipStream_t)stream, nullptr);
}
int b4r_ffn_block_bwd_marked(const b4r_ffn_desc* d, hipStream_t stream, hipEvent_t after_dx) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_ffn_block_bwd: null descriptor");
  B4R_CHECK_ARG(b4r_ffn_block_supported(d->H, d->I), B4R_E_SHAPE,
                "b4r_ffn_block_bwd: needs hidden size 64, inner size 256 and the bf16x3 mode (H=%d I=%d)", d->H, d->I);
  const bool slotmode = d->slot_positions != nullptr;
  const bool rowmode = d->rows != nullptr || slotmode;
  B4R_CHECK_ARG(slot_mode_ok(d), B4R_E_BADARG,
                "b4r_ffn_block_bwd: the slot mode needs slot_positions, slot_ids, slots_per_seq, seq_len, max_rows and no explicit list");
  B4R_CHECK_ARG(d->N > 0 && x1_given(d) && d->W1 && d->b1 && d->W2 && (d->dz2 || rowmode) && d->z1 && d->mean1 && d->rstd1 && d->ln1_gamma &&
                    d->dz1 && d->dW1 && d->db1 && d->dW2 && d->db2 && d->dln1_gamma && d->scratch,
                B4R_E_BADARG, "b4r_ffn_block_bwd: null argument");
  B4R_CHECK_ARG(!rowmode || ((slotmode || (d->n_rows && d->row_slot)) && d->max_rows > 0 && d->slot_grad && d->z2 && d->mean2 && d->rstd2 && d->ln_gamma &&
                             d->dln_gamma && d->dz2_rows),
                B4R_E_BADARG, "b4r_ffn_block_bwd: the row-list mode needs n_rows, max_rows, row_slot, slot_grad, z2, mean2, rstd2, "
                "ln_gamma, dln_gamma and dz2_rows");
  B4R_CHECK_ARG(al16(d->x1) && al16(d->W1) && al16(d->W2) && al16(d->dz2) && al16(d->z1) && al16(d->ln1_gamma) && al16(d->dz1) &&
                    al16(d->scratch),
                B4R_E_ALIGN, "b4r_ffn_block_bwd: operands must be 16-byte aligned");
  FfnP p = make_p(d);
  hipStream_t s = (hipStream_t)stream;
  const int units = rowmode ? d->max_rows : d->N;
  // the weight-gradient grid: the fewest workgroups (= partial slabs) that keep the longest workgroup's chunk count
  const int chunks = b4r_cdiv(units, CH_TOK), per_wg = b4r_cdiv(chunks, ffn_grid(chunks));
  const int gdx = ffn_grid(b4r_cdiv(units, 16)), gdw = b4r_cdiv(chunks, per_wg);
  float* sc = d->scratch;
  p.slab_w1 = sc; sc += (int64_t)gdw * HID * INNER;
  p.slab_w2 = sc; sc += (int64_t)gdw * HID * INNER;
  p.slab_b1 = sc; sc += (int64_t)gdw * INNER;
  p.slab_b2 = sc; sc += (int64_t)gdw * HID;
  p.ln_part = sc; sc += (int64_t)gdx * 128;
  p.ln2_part = sc;
  if (rowmode) { p.slotof = d->row_slot; p.dgr = d->slot_grad; p.dz2c = d->dz2_rows; }
  // measured (tools/bench_ffn.py, same box, dx + dw + reductions): 16 waves 81.2 / 81.4 us, 8 waves 84.0 / 85.7 us -- the spill-free
  // 256-register form loses more latency hiding (two waves per SIMD instead of four) than the 10 spilled registers cost
  static const int dx_waves = getenv("B4R_FFN_DX_WAVES") ? atoi(getenv("B4R_FFN_DX_WAVES")) : 16;
  int rc;
  if (dx_waves == 16) {
    rc = b4r_raise_lds((const void*)ffn_bwd_dx_kernel<16>, DX_LDS, "b4r_ffn_block_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(ffn_bwd_dx_kernel<16>, dim3(gdx), dim3(64 * 16), DX_LDS, s, p);
  } else {
    rc = b4r_raise_lds((const void*)ffn_bwd_dx_kernel<8>, DX_LDS, "b4r_ffn_block_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(ffn_bwd_dx_kernel<8>, dim3(gdx), dim3(64 * 8), DX_LDS, s, p);
  }
  B4R_CHECK_LAUNCH("b4r_ffn_block_bwd (dx)");
  if (after_dx != nullptr && hipEventRecord(after_dx, s) != hipSuccess) {
    b4r_set_error("b4r_ffn_block_bwd: event record failed");
    return B4R_E_HIP;
  }
  hipLaunchKernelGGL(ffn_bwd_dw_kernel, dim3(gdw), dim3(64 * FW), 0, s, p);
  B4R_CHECK_LAUNCH("b4r_ffn_block_bwd (dw)");
  // ordered sums over the workgroups (queued when the caller collects its reductions into one launch)
  rc = b4r_launch_slab_reduce_full(p.slab_w1, gdw, HID, INNER, d->dW1, INNER, 0, p.slab_b1, d->db1, nullptr, nullptr, s);
  if (rc) return rc;
  rc = b4r_launch_slab_reduce_full(p.slab_w2, gdw, INNER, HID, d->dW2, HID, 0, p.slab_b2, d->db2, nullptr, nullptr, s);
  if (rc) return rc;
  rc = b4r_launch_slab_reduce_full(p.ln_part, gdx, 1, 128, d->dln1_gamma, 128, 0, nullptr, nullptr, nullptr, nullptr, s);
  if (rc || !rowmode) return rc;
  return b4r_launch_slab_reduce_full(p.ln2_part, gdx, 1, 128, d->dln_gamma, 128, 0, nullptr, nullptr, nullptr, nullptr, s);
}
