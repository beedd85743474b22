// Evaluation path: candidate scores, stable descending ranking, rank of the ground truth, metric sums.
//
// Replaces  BERT4RecModel.rank_items          bert4rec/models/bert4rec_model.py:203-240   (gather candidate logits, tf.argsort
//                                             DESCENDING = ties keep the lower index, gather the candidates)
//           the rank lookup                   bert4rec/evaluation/bert4rec_evaluator.py:113-117
//           the metric accumulation           bert4rec/evaluation/evaluation_metrics.py:47-112
//
// Contract (DESIGN.md "ranking kernel", bit-exact against oracle/rank_oracle.c):
//   score(r, j) = fmaf-chain over k = 0 .. H-1 of hidden[r][k] * table[cand[r][j]][k], + bias[cand[r][j]]      (fp32)
//   pos(r, j)   = #{i : s_i > s_j} + #{i < j : s_i == s_j} ;  ranking[r][pos] = cand[r][j] ;  gt_rank = 1 + min pos of gt
// The reference computes ALL B*P*V logits to read 101 of them per user; here only the scores asked for are formed.
//
// Two paths:
//   * C <= RANK_LDS_MAX candidates per row (the evaluator's 101): one workgroup per row.  Candidate rows of the table are
//     fetched 256 at a time with full-line 16-byte loads (16 lanes per 256-byte row at H = 64: coalesced HBM gathers) into an
//     LDS tile with an odd row stride, then every thread walks ONE row with the k-ordered fma chain the contract prescribes;
//     scores stay in LDS, positions by counting.
//   * larger C, in particular cand == NULL = the whole vocabulary 0 .. C-1 (rank_items(items=None), bert4rec_model.py:236;
//     Reddit has 335 423 items): scores to global memory by the same tile kernel on a (row, tile) grid, then a stable LSD
//     radix argsort per row (4 passes of 8 bits on the order-preserving integer image of the score, descending; one
//     workgroup per row, a wave owns a contiguous chunk so that equal keys keep their index order).
#include <algorithm>

#include "b4r_common.h"

namespace {

constexpr int RT = 256;               // threads per workgroup = candidates per tile
constexpr int RANK_LDS_MAX = 8192;    // candidates per row the one-workgroup path keeps in LDS

__device__ __forceinline__ int64_t cand_at(const int64_t* cand, int64_t r, int C, int j) { return cand ? cand[r * C + j] : (int64_t)j; }

// scores of candidates j0 .. j0 + n - 1 of row r into sc[0..n) (LDS or global); tile: [tc][H + 1] floats; sh: hidden row [H]
// A candidate id outside [0, V) (the device sampler's -1 for a row that ran out of drawable items, or a caller's id beyond the
// vocabulary) scores -inf and loads nothing: the reference raises ValueError there (popular_random_sampler.py:104-109), the
// evaluator reports it after the batch -- the kernel must not read in front of / behind the table meanwhile.
__device__ __forceinline__ void score_tile(const float* sh, float* tile, const float* table, const float* bias, int H, int V,
                                           const int64_t* cand, int64_t r, int C, int j0, int n, int tc, float* sc,
                                           float* scores_out) {
  const int tid = threadIdx.x;
  const int h4 = H >> 2;
  for (int base = 0; base < n; base += tc) {
    const int m = min(tc, n - base);
    __syncthreads();   // the previous tile has been consumed
    for (int f = tid; f < m * h4; f += RT) {
      const int row = f / h4, c4 = f - row * h4;
      const int64_t c = cand_at(cand, r, C, j0 + base + row);
      const f32x4 v = (c >= 0 && c < V) ? *reinterpret_cast<const f32x4*>(table + c * H + 4 * c4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      float* dst = tile + row * (H + 1) + 4 * c4;
      dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    __syncthreads();
    if (tid < m) {
      const float* e = tile + tid * (H + 1);
      float acc = 0.f;
      for (int k = 0; k < H; ++k) acc = __builtin_fmaf(sh[k], e[k], acc);   // k-ordered fp32 fma chain (the contract)
      const int j = j0 + base + tid;
      const int64_t cj = cand_at(cand, r, C, j);
      const float s = (cj >= 0 && cj < V) ? acc + bias[cj] : -INFINITY;
      sc[j - j0] = s;
      if (scores_out) scores_out[r * (int64_t)C + j] = s;
    }
  }
}

__device__ __forceinline__ int tile_rows(int H) { return min(RT, 16384 / H); }

// ---- path 1: one workgroup per row, scores in LDS ------------------------------------------------------------------------
__global__ __launch_bounds__(RT) void rank_row_kernel(const float* hidden, int hidden_ld, const int64_t* hidden_row,
                                                      const float* table, const float* bias, int H, int V, const int64_t* cand,
                                                      int C, const int64_t* gt, int64_t* ranking, int32_t* gt_rank,
                                                      float* scores_out) {
  extern __shared__ float sm_rank[];   // [H] hidden row | [C] scores | tile
  __shared__ int s_best;
  float* sh = sm_rank; float* sc = sm_rank + H; float* tile = sc + C;
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t hr = hidden_row ? hidden_row[r] : r;
  for (int k = tid; k < H; k += RT) sh[k] = hidden[hr * hidden_ld + k];
  if (tid == 0) s_best = 0x7fffffff;
  score_tile(sh, tile, table, bias, H, V, cand, r, C, 0, C, tile_rows(H), sc, scores_out);
  __syncthreads();
  const int64_t g = gt ? gt[r] : -1;
  for (int j = tid; j < C; j += RT) {
    const float sj = sc[j];
    int pos = 0;
    for (int i = 0; i < C; ++i) {
      const float si = sc[i];
      pos += (si > sj || (si == sj && i < j)) ? 1 : 0;
    }
    const int64_t cj = cand_at(cand, r, C, j);
    if (ranking) ranking[r * (int64_t)C + pos] = cj;
    if (gt && cj == g) atomicMin(&s_best, pos);
  }
  __syncthreads();
  if (tid == 0 && gt_rank) gt_rank[r] = (s_best == 0x7fffffff) ? 0 : s_best + 1;
}

// ---- path 2a: scores of one (row, 256-candidate tile) to global memory ------------------------------------------------
__global__ __launch_bounds__(RT) void rank_scores_kernel(const float* hidden, int hidden_ld, const int64_t* hidden_row,
                                                         const float* table, const float* bias, int H, int V, const int64_t* cand,
                                                         int C, int64_t r0, float* scores /* [rows][C] */) {
  extern __shared__ float sm_rank[];   // [H] | tile
  float* sh = sm_rank; float* tile = sm_rank + H;
  const int64_t r = r0 + blockIdx.y;
  const int64_t hr = hidden_row ? hidden_row[r] : r;
  for (int k = threadIdx.x; k < H; k += RT) sh[k] = hidden[hr * hidden_ld + k];
  const int j0 = blockIdx.x * RT, n = min(RT, C - j0);
  // scores_out indexing inside score_tile is r * C + j with the GLOBAL row; offset the base so that row r0 lands on row 0
  score_tile(sh, tile, table, bias, H, V, cand, r, C, j0, n, tile_rows(H), scores + (blockIdx.y * (int64_t)C + j0), nullptr);
}

// ---- path 2b: stable descending argsort of each row's scores ----------------------------------------------------------
// order-preserving image of a float for an ASCENDING unsigned sort that yields DESCENDING scores; -0.0 counts as +0.0 (the
// comparison the contract uses makes them equal)
__device__ __forceinline__ uint32_t desc_key(float s) {
  uint32_t u = __builtin_bit_cast(uint32_t, s == 0.f ? 0.f : s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending image
  return ~u;                                        // descending
}

constexpr int SW = 16;   // waves of the sorting workgroup

// one workgroup per row; keys / values ping-pong between (k0, v0) and (k1, v1), each [rows][C]; the result ends in (k0, v0)
__global__ __launch_bounds__(64 * SW) void rank_argsort_kernel(const float* scores, int C, uint32_t* k0, uint32_t* v0,
                                                              uint32_t* k1, uint32_t* v1) {
  __shared__ uint32_t hist[SW][256];
  __shared__ uint32_t tot[256];
  const int64_t row = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* s = scores + row * C;
  uint32_t* ka = k0 + row * C; uint32_t* va = v0 + row * C;
  uint32_t* kb = k1 + row * C; uint32_t* vb = v1 + row * C;
  for (int j = threadIdx.x; j < C; j += 64 * SW) { ka[j] = desc_key(s[j]); va[j] = (uint32_t)j; }
  const int chunk = (((C + SW - 1) / SW) + 63) & ~63;   // elements per wave, a multiple of 64
  const int begin = min(C, w * chunk), end = min(C, begin + chunk);
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    for (int d = lane; d < 256; d += 64) hist[w][d] = 0;
    __threadfence_block();
    __syncthreads();   // also: the previous pass' (or the initial) global writes of this workgroup are visible
    for (int j = begin + lane; j < end; j += 64) atomicAdd(&hist[w][(ka[j] >> shift) & 255u], 1u);
    __syncthreads();
    // exclusive offsets: digit-major, wave-minor
    if (threadIdx.x < 256) {
      uint32_t run = 0;
      for (int ww = 0; ww < SW; ++ww) { const uint32_t c = hist[ww][threadIdx.x]; hist[ww][threadIdx.x] = run; run += c; }
      tot[threadIdx.x] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int d = 0; d < 256; ++d) { const uint32_t c = tot[d]; tot[d] = run; run += c; }
    }
    __syncthreads();
    // scatter, 64 elements of the wave's chunk at a time, in order: rank among the equal digits of the group by ballots
    for (int j0 = begin; j0 < end; j0 += 64) {
      const int j = j0 + lane;
      const bool live = j < end;
      const uint32_t key = live ? ka[j] : 0u, val = live ? va[j] : 0u;
      const uint32_t d = (key >> shift) & 255u;
      uint64_t same = __ballot(live);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint64_t m = __ballot((d >> b) & 1u);
        same &= ((d >> b) & 1u) ? m : ~m;
      }
      const uint32_t before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
      const uint32_t count = (uint32_t)__popcll(same);
      uint32_t base = 0;
      if (live) base = tot[d] + hist[w][d];
      // every lane of a digit group reads the same base before its leader bumps it
      __builtin_amdgcn_wave_barrier();
      if (live && before == 0) hist[w][d] += count;
      __builtin_amdgcn_wave_barrier();
      if (live) { kb[base + before] = key; vb[base + before] = val; }
    }
    __threadfence_block();
    __syncthreads();
    uint32_t* t = ka; ka = kb; kb = t;
    t = va; va = vb; vb = t;
  }
  // 4 passes: the result is back in (k0, v0)
}

// ranking[r][pos] = candidate id of the value at pos; gt_rank from the position of the ground truth
__global__ __launch_bounds__(256) void rank_emit_kernel(const uint32_t* v0, const int64_t* cand, int C, int64_t r0,
                                                        const int64_t* gt, int64_t* ranking, int32_t* gt_rank) {
  const int64_t lr = blockIdx.y, r = r0 + lr;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= C) return;
  const uint32_t j = v0[lr * C + pos];
  const int64_t cj = cand_at(cand, r, C, (int)j);
  if (ranking) ranking[r * (int64_t)C + pos] = cj;
  if (gt_rank && gt && cj == gt[r]) atomicMin(reinterpret_cast<int*>(gt_rank) + r, pos + 1);
}
__global__ void rank_init_gt_kernel(int32_t* gt_rank, int64_t r0, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) gt_rank[r0 + i] = 0x7fffffff;
}
__global__ void rank_fix_gt_kernel(int32_t* gt_rank, int64_t r0, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n && gt_rank[r0 + i] == 0x7fffffff) gt_rank[r0 + i] = 0;
}

// ---- metric sums ------------------------------------------------------------------------------------------------------------
// families as bert4rec_amd/evaluation/evaluation_metrics.py: 0 count, 1 hit@k, 2 ndcg@k, 3 reciprocal rank.
// sums[m] += sum over ranks > 0 of gain_m(rank); users[0] += number of ranks > 0.  One workgroup, fixed summation order.
constexpr int MAX_METRICS = 32;
struct MetricP { int family[MAX_METRICS]; int cutoff[MAX_METRICS]; int n; };
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int32_t* ranks, int R, MetricP mp, double* sums, int64_t* users) {
  __shared__ double red[256];
  __shared__ int64_t cnt[256];
  const int tid = threadIdx.x;
  int64_t c = 0;
  for (int i = tid; i < R; i += 256) c += ranks[i] > 0 ? 1 : 0;
  cnt[tid] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) cnt[tid] += cnt[tid + o]; __syncthreads(); }
  if (tid == 0) users[0] += cnt[0];
  for (int m = 0; m < mp.n; ++m) {
    const int fam = mp.family[m], k = mp.cutoff[m];
    double a = 0.0;
    for (int i = tid; i < R; i += 256) {
      const int rk = ranks[i];
      if (rk <= 0) continue;
      double g = 0.0;
      if (fam == 0) g = 1.0;
      else if (fam == 1) g = rk <= k ? 1.0 : 0.0;
      else if (fam == 2) g = rk <= k ? (rk == 1 ? 1.0 : 1.0 / log2((double)rk + 1.0)) : 0.0;
      else if (fam == 3) g = 1.0 / (double)rk;
      a += g;
    }
    __syncthreads();
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) sums[m] += red[0];
  }
}

}  // namespace

extern "C" int64_t b4r_rank_scratch_bytes(int32_t R, int32_t C) {
  if (R <= 0 || C <= RANK_LDS_MAX) return 0;
  return (int64_t)R * C * 20;   // scores + two (key, index) buffers; fewer bytes are accepted: rows are then ranked in groups
}

extern "C" int b4r_rank_candidates(const float* hidden, int32_t hidden_ld, const int64_t* hidden_row, const float* table,
                                   const float* bias, int32_t H, int32_t V, const int64_t* cand, int32_t R, int32_t C,
                                   const int64_t* gt, int64_t* ranking, int32_t* gt_rank, float* scores, void* scratch,
                                   int64_t scratch_bytes, b4r_stream_t stream) {
  B4R_CHECK_ARG(hidden && table && bias, B4R_E_BADARG, "b4r_rank_candidates: null argument");
  B4R_CHECK_ARG(R > 0 && C > 0 && H > 0 && H % 4 == 0 && H <= 4096 && hidden_ld >= H && V > 0, B4R_E_SHAPE, "b4r_rank_candidates: bad shape");
  B4R_CHECK_ARG(cand != nullptr || C <= V, B4R_E_SHAPE, "b4r_rank_candidates: cand == NULL ranks items 0 .. C-1, C = %d exceeds the table's %d rows", C, V);
  B4R_CHECK_ARG(b4r_aligned16(table), B4R_E_ALIGN, "b4r_rank_candidates: the table must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int tc = std::min(RT, 16384 / H);
  const size_t tile = (size_t)tc * (H + 1) * sizeof(float);
  if (C <= RANK_LDS_MAX) {
    const size_t sh = (size_t)(H + C) * sizeof(float) + tile;
    int rc = b4r_raise_lds((const void*)rank_row_kernel, sh, "b4r_rank_candidates");
    if (rc) return rc;
    hipLaunchKernelGGL(rank_row_kernel, dim3(R), dim3(RT), sh, s, hidden, hidden_ld, hidden_row, table, bias, H, V, cand, C, gt,
                       ranking, gt_rank, scores);
    B4R_CHECK_LAUNCH("b4r_rank_candidates");
    return B4R_OK;
  }
  const int64_t per_row = (int64_t)C * 20;
  B4R_CHECK_ARG(scratch && scratch_bytes >= per_row && b4r_aligned16(scratch), B4R_E_NOMEM,
                "b4r_rank_candidates: %d candidates per row need a scratch of at least %lld bytes (b4r_rank_scratch_bytes)", C,
                (long long)per_row);
  const int64_t group = std::min<int64_t>(R, std::min<int64_t>(scratch_bytes / per_row, 65535));
  const size_t sh = (size_t)H * sizeof(float) + tile;
  int rc = b4r_raise_lds((const void*)rank_scores_kernel, sh, "b4r_rank_candidates");
  if (rc) return rc;
  float* sc = static_cast<float*>(scratch);
  uint32_t* k0 = reinterpret_cast<uint32_t*>(sc + group * C);
  uint32_t* v0 = k0 + group * C;
  uint32_t* k1 = v0 + group * C;
  uint32_t* v1 = k1 + group * C;
  for (int64_t r0 = 0; r0 < R; r0 += group) {
    const int n = (int)std::min<int64_t>(group, R - r0);
    hipLaunchKernelGGL(rank_scores_kernel, dim3(b4r_cdiv(C, RT), n), dim3(RT), sh, s, hidden, hidden_ld, hidden_row, table, bias, H, V,
                       cand, C, r0, sc);
    if (scores) {
      if (hipMemcpyAsync(scores + r0 * C, sc, (size_t)n * C * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) {
        b4r_set_error("b4r_rank_candidates: copying the scores failed");
        return B4R_E_HIP;
      }
    }
    if (ranking || gt_rank) {
      hipLaunchKernelGGL(rank_argsort_kernel, dim3(n), dim3(64 * SW), 0, s, sc, C, k0, v0, k1, v1);
      if (gt_rank && gt) hipLaunchKernelGGL(rank_init_gt_kernel, dim3(b4r_cdiv(n, 256)), dim3(256), 0, s, gt_rank, r0, n);
      hipLaunchKernelGGL(rank_emit_kernel, dim3(b4r_cdiv(C, 256), n), dim3(256), 0, s, v0, cand, C, r0, gt, ranking, gt_rank);
      if (gt_rank && gt) hipLaunchKernelGGL(rank_fix_gt_kernel, dim3(b4r_cdiv(n, 256)), dim3(256), 0, s, gt_rank, r0, n);
    }
  }
  B4R_CHECK_LAUNCH("b4r_rank_candidates (whole-vocabulary path)");
  return B4R_OK;
}

extern "C" int b4r_rank_metrics(const int32_t* gt_rank, int32_t R, const int32_t* family, const int32_t* cutoff, int32_t n_metrics,
                                double* gain_sums, int64_t* users, b4r_stream_t stream) {
  B4R_CHECK_ARG(gt_rank && family && cutoff && gain_sums && users, B4R_E_BADARG, "b4r_rank_metrics: null argument");
  B4R_CHECK_ARG(R > 0 && n_metrics > 0 && n_metrics <= MAX_METRICS, B4R_E_SHAPE, "b4r_rank_metrics: 1..%d metrics", MAX_METRICS);
  MetricP mp{};
  mp.n = n_metrics;
  for (int m = 0; m < n_metrics; ++m) {
    B4R_CHECK_ARG(family[m] >= 0 && family[m] <= 3, B4R_E_BADARG, "b4r_rank_metrics: unknown gain family %d", family[m]);
    mp.family[m] = family[m]; mp.cutoff[m] = cutoff[m];
  }
  hipLaunchKernelGGL(rank_metrics_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gt_rank, R, mp, gain_sums, users);
  B4R_CHECK_LAUNCH("b4r_rank_metrics");
  return B4R_OK;
}
