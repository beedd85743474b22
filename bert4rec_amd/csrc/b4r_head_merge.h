// Merge of the V slices of the logits-free masked-LM head's forward (b4r_head_rx.hip: head_fwd_kernel leaves, per slice and row,
// H accumulators of sum_v exp(x - max) E[v,:], the slice's max and sum in log2 units, its best logit and index): flash-decoding
// style.  Shared by head_combine_kernel (the forward's second launch), head_dE_kernel (lse / labels of the rows it sweeps when the
// merge is deferred) and the LayerNorm backward of the transform (ln_bwd_kernel<.., MERGE>: in a train step the merged dT is consumed
// right there, so the merge needs no launch and dT no round trip).
#pragma once
#include "b4r_common.h"

namespace {

constexpr int part_ld(int nkh) { return 32 * nkh + 8; }   // floats per (V slice, row): H accumulators, max, sum, best logit, best index
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

struct HeadMergeP {
  const float* part; int slices, M, V;
  const float* T; const float* E; const float* bias; const int64_t* y;
  float* row_out; float* lse_out; int32_t* ylab;   // [M,4] loss rows as b4r_softmax_ce writes them, [M], [M]
};

// max and sum of a row over the forward's V slices (at most CMAX: the host does not fold the merge into dE beyond that), requested
// early and finished a chunk later -> log-sum-exp in natural units (+inf for a slot without a label: zero gradient rows in
// head_dE_kernel) and the label (-1: none).  The same arithmetic as head_merge_row: slices past the end add +0.
#ifndef B4R_CMAX
#define B4R_CMAX 16   // (the 32 x 32-tile forward uses up to 16 slices: Steam 12; 8 until round 4)
#endif
constexpr int CMAX = B4R_CMAX;
struct RowPart { float mx[CMAX], sm[CMAX]; long long y; };
// ms: the forward's compact copy [slices][M][2] of (max, sum), behind its records
__device__ __forceinline__ void row_part_fetch(RowPart& rp, const float* ms, int slices, int M, const int64_t* y, int m) {
#pragma unroll
  for (int s = 0; s < CMAX; ++s) {
    rp.mx[s] = -INFINITY; rp.sm[s] = 0.f;
    if (s < slices) {
      const float* src = ms + ((int64_t)s * M + m) * 2;
      rp.mx[s] = src[0]; rp.sm[s] = src[1];
    }
  }
  rp.y = y[m];
}
__device__ __forceinline__ void row_part_finish(const RowPart& rp, int V, float& lse_out, int& lab_out) {
  float mx = -INFINITY;
#pragma unroll
  for (int s = 0; s < CMAX; ++s) mx = fmaxf(mx, rp.mx[s]);
  float sum = 0.f;
#pragma unroll
  for (int s = 0; s < CMAX; ++s) sum += rp.sm[s] * ex2(rp.mx[s] - mx);
  const bool valid = (rp.y != 0), y_ok = (rp.y >= 0 && rp.y < V);
  lse_out = valid ? (mx + __log2f(sum)) * LN2 : INFINITY;
  lab_out = (valid && y_ok) ? (int)rp.y : -1;
}

// merge the V slices of row m; H/4 threads per row (4 columns each: c4), whole TPR-lane groups call this together.  Returns this
// thread's four columns of dT = softmax - onehot (zero for a slot without a label); lane c4 == 0 writes the row's scalars.
template <int NKH>
__device__ __forceinline__ f32x4 head_merge_row(const HeadMergeP& q, int m, int c4) {
  const float* part = q.part; const int slices = q.slices, M = q.M, V = q.V;
  const float* T = q.T; const float* E = q.E; const float* bias = q.bias; const int64_t* y = q.y;
  float* row_out = q.row_out; float* lse_out = q.lse_out; int32_t* ylab = q.ylab;
  constexpr int H = 32 * NKH, TPR = 8 * NKH, PART_LD = part_ld(NKH);   // TPR = threads per row: 16, 32 or 64
  float mx = -INFINITY;
#pragma unroll 4
  for (int s = 0; s < slices; ++s) mx = fmaxf(mx, part[((int64_t)s * M + m) * PART_LD + H]);
  float sum = 0.f, best = -INFINITY;
  int bidx = 0x7fffffff;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int s = 0; s < slices; ++s) {                            // slices in increasing column order: lowest index wins ties
    const float* src = part + ((int64_t)s * M + m) * PART_LD;
    const float w = ex2(src[H] - mx);                           // the sweep's maxima are in log2 units
    sum += src[H + 1] * w;
    acc += *reinterpret_cast<const f32x4*>(src + 4 * c4) * w;
    if (src[H + 2] > best) { best = src[H + 2]; bidx = __float_as_int(src[H + 3]); }
  }
  const int64_t y64 = y[m];
  const bool valid = (y64 != 0), y_ok = (y64 >= 0 && y64 < V);
  f32x4 d = {0.f, 0.f, 0.f, 0.f}, ey = {0.f, 0.f, 0.f, 0.f};
  if (y_ok) ey = *reinterpret_cast<const f32x4*>(E + y64 * H + 4 * c4);
  if (valid) d = acc * (1.0f / sum) - ey;
  // the label's logit in plain fp32 (the loss needs its value, the metrics only the argmax index): TPR lanes x 4 columns
  const f32x4 tv = *reinterpret_cast<const f32x4*>(T + (int64_t)m * H + 4 * c4);
  float xl = (tv[0] * ey[0] + tv[1] * ey[1]) + (tv[2] * ey[2] + tv[3] * ey[3]);
#pragma unroll
  for (int o = 1; o < TPR; o <<= 1) xl += __shfl_xor(xl, o, 64);
  if (c4 == 0) {
    if (y_ok) xl += bias[y64];
    const float lse = (mx + __log2f(sum)) * LN2;
    row_out[4 * (int64_t)m + 0] = (valid && y_ok) ? (lse - xl) : 0.f;
    row_out[4 * (int64_t)m + 1] = valid ? 1.f : 0.f;
    row_out[4 * (int64_t)m + 2] = (valid && (int64_t)bidx == y64) ? 1.f : 0.f;
    row_out[4 * (int64_t)m + 3] = ((int64_t)bidx == y64) ? 1.f : 0.f;
    lse_out[m] = valid ? lse : INFINITY;                        // +inf => zero gradient rows in head_dE_kernel
    ylab[m] = (valid && y_ok) ? (int32_t)y64 : -1;
  }
  return d;
}

}  // namespace
