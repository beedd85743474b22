// Masked-LM head of a TRAIN step without materialising the [M, V] logits (split-precision arithmetic; hidden size 64, 128
// or 256 = NKH column groups of 32, the kernels are templates over NKH).
//
//   logits x[m,v] = T[m,:].E[v,:] + b[v]      (T = transform output [M,64], E = tied item table [V,64])
//   loss_m = logsumexp_v x[m,v] - x[m,y_m] ;  g[m,v] = softmax(x[m,:])[v] - [v == y_m]   (rows with y_m == 0 ignored)
//   dT = g.E ;  dE = g^T.T ;  db = column sums of g
//
// The materialising path (b4r_gemm_rx.hip + softmax_ce_kernel + two more passes over the 152 MB of dlogits at ML-1M)
// moves ~760 MB per step through HBM; here the logits only ever exist as accumulator tiles:
//   head_fwd_kernel   "attention" over the vocabulary (b4r_attn_rx.hip's forward with E as both keys and values): a wave
//                     owns 16 rows of T, sweeps a slice of V in 16-row tiles of E held in LDS as bf16 hi / lo images,
//                     keeps a running max / sum (online softmax) and accumulates sum_v exp(x - max) E[v,:] with the
//                     probability tiles fed back as MFMA operands; also argmax and the label logit for the metrics.
//   head_combine_kernel merges the V slices of a row (flash-decoding style), writes loss rows, lse and dT = acc/sum - E[y].
//   head_dE_kernel    the other orientation (a wave owns 16 rows of E, sweeps slices of M): recomputes the logit tiles
//                     with the same arithmetic, g = exp(x - lse) - onehot, dE^T += T^T.g, db += column sums; partial
//                     tiles per M slice go to slabs that the backward's deferred ordered reduction sums.
// Tile / image mechanics are those of b4r_rx_tiles.h: the H columns of E (or T) are NKH 32-column images (hi, lo each),
// interleaved per 16-row tile.
#include "b4r_rx_tiles.h"
#include "b4r_head_merge.h"
#ifndef HEAD_CH2
#define HEAD_CH2 10
#endif

namespace {

// 16-row tiles per LDS chunk (even): 40 KB of images at H = 64 (three workgroups per CU), 48 KB at 128, 64 KB at 256
constexpr int head_ch(int nkh) { return nkh == 2 ? HEAD_CH2 : nkh == 4 ? 6 : 4; }
// (part_ld, LOG2E / LN2 and ex2 -- the sweeps work in log2 units, T or E is scaled by log2(e) before it is split so that the softmax
// exponential is the bare v_exp_f32 -- are in b4r_head_merge.h, shared with the LayerNorm backward that can do the merge)

struct HeadP {
  const float* T; const float* E; const float* bias; const int64_t* y;
  float* part;                    // forward: [slices][M][PART_LD]
  const float* lse; const int32_t* ylab;
  float* slab; float* bslab;      // dE: [slices][V][64], [slices][V]
  int M, V;
  int tiles_per_slice;            // even number of 16-row tiles per slice (of V in the forward, of M in dE)
  // dE in front of the merge of the forward's V slices (cpart != NULL: the forward ran its sweep only, the LayerNorm backward behind
  // this launch merges: b4r_head_merge.h): every workgroup forms the lse / label of the rows it sweeps from the forward's partials
  // itself (head_merge_row's arithmetic, the same bits)
  const float* cpart; int cslices;
};

// image index as a function argument: it is a constant after unrolling, so the offset still folds into the instruction
__device__ __forceinline__ bf16x8 row_frag_at(const char* tile, int img) {
  return *reinterpret_cast<const bf16x8*>(tile + img * IMG_BYTES);
}
template <int TB>
__device__ __forceinline__ bf16x8 tr_frag_at(const char* tile, int img) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + img * IMG_BYTES));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + img * IMG_BYTES + TB));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NKH>
__device__ __forceinline__ f32x4 logit_tile(const char* tile, const bf16x8 (&bh)[NKH], const bf16x8 (&bl)[NKH]) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kh = 0; kh < NKH; ++kh) s = mfma3(row_frag_at(tile, 2 * kh), row_frag_at(tile, 2 * kh + 1), bh[kh], bl[kh], s);
  return s;
}

// acc[2*kh + db] += image(kh)^T[16 columns of block db][rows of the tile pair] . (ph, pl)
template <int NKH>
__device__ __forceinline__ void feed_pair(const char* img, const FragAddr& fa, int pair_tile, const bf16x8 ph, const bf16x8 pl,
                                          f32x4 (&acc)[2 * NKH]) {
  constexpr int TB = 2 * NKH * IMG_BYTES;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const char* tile = img + fa.tr[db] + TB * pair_tile;
#pragma unroll
    for (int kh = 0; kh < NKH; ++kh)
      acc[2 * kh + db] = mfma3(tr_frag_at<TB>(tile, 2 * kh), tr_frag_at<TB>(tile, 2 * kh + 1), ph, pl, acc[2 * kh + db]);
  }
}

// one chunk of `nrows` rows (row c0 onwards of a [*, 32*NKH] fp32 matrix) -> the chunk's images; rows >= valid are zero.
// Split in two so that the loads of chunk c+1 are in flight while chunk c is multiplied (a chunk is only 5 tile pairs of
// work; waiting out the load latency at every chunk boundary was a quarter of the kernel).
template <int NKH> struct ChunkRegs {
  static constexpr int NIT = (head_ch(NKH) * 16 * 8 + 64 * WAVES - 1) / (64 * WAVES);   // float4 pieces per thread and 32-column group
  StagedRowsT<NIT> st[NKH / 2];
};
template <int NKH>
__device__ __forceinline__ void chunk_fetch(ChunkRegs<NKH>& r, const float* src, int c0, int valid) {
  constexpr int H = 32 * NKH;
#pragma unroll
  for (int j = 0; j < NKH / 2; ++j)
    stage_fetch<ChunkRegs<NKH>::NIT>(r.st[j], src + (int64_t)c0 * H + 64 * j, H, src + (int64_t)c0 * H + 64 * j + 32, H, 0, valid);
}
template <int NKH>
__device__ __forceinline__ void chunk_write(const ChunkRegs<NKH>& r, char* img, int nrows, int valid) {
  constexpr int TB = 2 * NKH * IMG_BYTES;
#pragma unroll
  for (int j = 0; j < NKH / 2; ++j) stage_write<ChunkRegs<NKH>::NIT, TB>(r.st[j], img + 4 * j * IMG_BYTES, nrows, valid);
}

// -----------------------------------------------------------------------------------------------------------
// forward: grid (row blocks of 128, V slices); wave = 16 rows of T x the slice's rows of E
// LDS: [E chunk images | bias chunk]
// -----------------------------------------------------------------------------------------------------------
template <int NKH>
__global__ __launch_bounds__(64 * WAVES) void head_fwd_kernel(HeadP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_head[];
  constexpr int H = 32 * NKH, TB = 2 * NKH * IMG_BYTES, HEAD_CH = head_ch(NKH), PART_LD = part_ld(NKH);
  char* img = smem_head;
  float* sBias = reinterpret_cast<float*>(img + HEAD_CH * TB);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const int m = blockIdx.x * ROWS_WG + 16 * wave + i;
  const int mc = min(m, p.M - 1);
  bf16x8 th[NKH], tl[NKH];
#pragma unroll
  for (int kh = 0; kh < NKH; ++kh) split8(load8(p.T + (int64_t)mc * H + 32 * kh + 8 * g) * LOG2E, th[kh], tl[kh]);
  const int v_begin = blockIdx.y * p.tiles_per_slice * 16;
  const int v_end = min(((p.V + 31) >> 5) << 5, v_begin + p.tiles_per_slice * 16);   // multiples of 32; v_begin < V
  const FragAddr fa = frag_addr(lane);

  float mx = -INFINITY, sum = 0.f, best = -INFINITY;   // log2 units
  int bidx = 0x7fffffff;
  f32x4 acc[2 * NKH];
#pragma unroll
  for (int kb = 0; kb < 2 * NKH; ++kb) acc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  ChunkRegs<NKH> regs;
  auto chunk_valid = [&](int c0) { return min(min(HEAD_CH * 16, v_end - c0), p.V - c0); };   // >= 1: chunks start below V
  chunk_fetch<NKH>(regs, p.E, v_begin, chunk_valid(v_begin));
  float bz = p.bias[min(v_begin + (int)threadIdx.x, p.V - 1)];
  for (int c0 = v_begin; c0 < v_end; c0 += HEAD_CH * 16) {
    const int nrows = min(HEAD_CH * 16, v_end - c0);            // multiple of 32
    __syncthreads();                                            // the previous chunk has been consumed
    chunk_write<NKH>(regs, img, nrows, chunk_valid(c0));
    if ((int)threadIdx.x < nrows) sBias[threadIdx.x] = (c0 + (int)threadIdx.x < p.V) ? bz * LOG2E : -INFINITY;
    __syncthreads();
    const int cn = c0 + HEAD_CH * 16;
    if (cn < v_end) {                                           // block-uniform: the next chunk travels during this one
      chunk_fetch<NKH>(regs, p.E, cn, chunk_valid(cn));
      bz = p.bias[min(cn + (int)threadIdx.x, p.V - 1)];
    }
    for (int tp = 0; tp < nrows / 32; ++tp) {
      f32x4 x[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
        x[u] = logit_tile<NKH>(img + fa.row + TB * (2 * tp + u), th, tl) +
               *reinterpret_cast<const f32x4*>(&sBias[16 * (2 * tp + u) + 4 * g]);
      const float pl8 = fmaxf(fmaxf(fmaxf(x[0][0], x[0][1]), fmaxf(x[0][2], x[0][3])), fmaxf(fmaxf(x[1][0], x[1][1]), fmaxf(x[1][2], x[1][3])));
      if (pl8 > best) {                                         // rare after the first tiles: argmax of this lane's columns
        best = pl8;
        int j = 7;                                              // lowest of the 8 positions that holds the maximum
#pragma unroll
        for (int jj = 6; jj >= 0; --jj) j = (x[jj >> 2][jj & 3] == pl8) ? jj : j;
        bidx = c0 + 16 * (2 * tp + (j >> 2)) + 4 * g + (j & 3);   // columns grow with the pair: earlier maxima win ties
      }
      // The reference `mx` of the running sums only has to be COMMON to the four lanes of a row and close enough to the
      // row maximum that 2^(x - mx) cannot overflow; it need not be the maximum.  So it is moved (to the row maximum so
      // far: two cross-lane exchanges, one rescale) only when some logit of the wave exceeds its row's reference by more
      // than 2^SLACK -- at the first pair of a slice (mx = -inf) and then almost never -- and the common path is one
      // compare and a wave-uniform branch.  The terms stay below 2^SLACK, their sum below 2^SLACK V: no overflow, and the
      // relative precision of a floating-point sum does not depend on the reference.
      constexpr float SLACK = 8.0f;
      if (__builtin_amdgcn_ballot_w64(pl8 > mx + SLACK) != 0) {
        float pm = fmaxf(pl8, __shfl_xor(pl8, 16, 64));
        pm = fmaxf(pm, __shfl_xor(pm, 32, 64));
        const float mnew = fmaxf(mx, pm);                       // finite: the first pair of a slice holds real columns
        const float alpha = (mx == mnew) ? 1.0f : ex2(mx - mnew);
        sum *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2 * NKH; ++kb) acc[kb] = acc[kb] * alpha;
        mx = mnew;
      }
      f32x4 pr[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = ex2(x[u][r] - mx);
          pr[u][r] = e;
          sum += e;
        }
      }
      bf16x8 ph, pl;
      split8(cat(pr[0], pr[1]), ph, pl);
      feed_pair<NKH>(img, fa, 2 * tp, ph, pl, acc);             // acc^T[k][row] += E^T[k][v pair] . p^T[v pair][row]
    }
  }
  // the four g lanes of a row hold disjoint columns: combine
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (m < p.M) {
    float* dst = p.part + ((int64_t)blockIdx.y * p.M + m) * PART_LD;
#pragma unroll
    for (int kb = 0; kb < 2 * NKH; ++kb) *reinterpret_cast<f32x4*>(dst + 16 * kb + 4 * g) = acc[kb];
    if (g == 0) {
      dst[H] = mx; dst[H + 1] = sum; dst[H + 2] = best; dst[H + 3] = __int_as_float(bidx);
      // (max, sum) once more as a compact [slices][M][2] array behind the records: what head_dE_kernel reads when it forms the lse
      // itself (one 8-byte load per slice and row, coalesced -- out of the records it was a cache line per slice and row, +4 us)
      float* ms = p.part + (int64_t)gridDim.y * p.M * PART_LD + ((int64_t)blockIdx.y * p.M + m) * 2;
      ms[0] = mx; ms[1] = sum;
    }
  }
}

template <int NKH>
__global__ __launch_bounds__(256) void head_combine_kernel(const float* part, int slices, int M, int V, const float* T,
                                                           const float* E, const float* bias, const int64_t* y, float* dT,
                                                           float* row_out, float* lse_out, int32_t* ylab) {
  constexpr int TPR = 8 * NKH;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx / TPR >= M) return;                                   // whole TPR-lane groups are in or out of range together
  constexpr int H = 32 * NKH;
  const HeadMergeP mp{part, slices, M, V, T, E, bias, y, row_out, lse_out, ylab};
  const f32x4 d = head_merge_row<NKH>(mp, idx / TPR, idx % TPR);
  *reinterpret_cast<f32x4*>(dT + (int64_t)(idx / TPR) * H + 4 * (idx % TPR)) = d;
}

// -----------------------------------------------------------------------------------------------------------
// dE / db: grid (blocks of 128 rows of E, M slices); wave = 16 rows of E x the slice's rows of T
// LDS: [T chunk images | lse chunk | label chunk]
// -----------------------------------------------------------------------------------------------------------
// FOLD: lse / labels from the forward's partials (HeadP::cpart) instead of the merged arrays
template <int NKH, bool FOLD>
__global__ __launch_bounds__(64 * WAVES) void head_dE_kernel(HeadP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_head[];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr int H = 32 * NKH, TB = 2 * NKH * IMG_BYTES, HEAD_CH = head_ch(NKH);
  char* img = smem_head;
  float* sLse = reinterpret_cast<float*>(img + HEAD_CH * TB);
  int* sY = reinterpret_cast<int*>(sLse + HEAD_CH * 16);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = lane & 15, g = lane >> 4;
  const int v = blockIdx.x * ROWS_WG + 16 * wave + i;
  const bool vlive = v < p.V;
  const int vc = min(v, p.V - 1);
  bf16x8 eh[NKH], el[NKH];
#pragma unroll
  for (int kh = 0; kh < NKH; ++kh) split8(load8(p.E + (int64_t)vc * H + 32 * kh + 8 * g) * LOG2E, eh[kh], el[kh]);
  const float bv = vlive ? p.bias[vc] * LOG2E : -INFINITY;      // -inf => zero probability, and no label equals v >= V
  const int m_begin = blockIdx.y * p.tiles_per_slice * 16;
  const int m_end = min(((p.M + 31) >> 5) << 5, m_begin + p.tiles_per_slice * 16);
  const FragAddr fa = frag_addr(lane);

  float dbsum = 0.f;
  f32x4 acc[2 * NKH];
#pragma unroll
  for (int kb = 0; kb < 2 * NKH; ++kb) acc[kb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  ChunkRegs<NKH> regs;
  auto chunk_valid = [&](int c0) { return min(min(HEAD_CH * 16, m_end - c0), p.M - c0); };
  chunk_fetch<NKH>(regs, p.T, m_begin, chunk_valid(m_begin));
  float lz = INFINITY;
  int yz = -1;
  RowPart rp;
  const bool row_thread = (int)threadIdx.x < HEAD_CH * 16;        // thread t: row t of the chunk
  auto row_request = [&](int m) __attribute__((always_inline)) {
    if (!row_thread) return;
    if (FOLD) row_part_fetch(rp, p.cpart + (int64_t)p.cslices * p.M * part_ld(NKH), p.cslices, p.M, p.y, m);
    else { lz = p.lse[m]; yz = p.ylab[m]; }
  };
  auto row_finish = [&]() __attribute__((always_inline)) {
    if (FOLD && row_thread) row_part_finish(rp, p.V, lz, yz);
  };
  row_request(min(m_begin + (int)threadIdx.x, p.M - 1));
  row_finish();
  for (int c0 = m_begin; c0 < m_end; c0 += HEAD_CH * 16) {
    const int nrows = min(HEAD_CH * 16, m_end - c0);
    __syncthreads();                                            // the previous chunk has been consumed
    chunk_write<NKH>(regs, img, nrows, chunk_valid(c0));
    if ((int)threadIdx.x < nrows) {
      const bool in = c0 + (int)threadIdx.x < p.M;
      sLse[threadIdx.x] = in ? lz * LOG2E : INFINITY;
      sY[threadIdx.x] = in ? yz : -1;
    }
    __syncthreads();
    const int cn = c0 + HEAD_CH * 16;
    if (cn < m_end) {                                           // block-uniform: the next chunk travels during this one
      chunk_fetch<NKH>(regs, p.T, cn, chunk_valid(cn));
      row_request(min(cn + (int)threadIdx.x, p.M - 1));
    }
    for (int tp = 0; tp < nrows / 32; ++tp) {
      f32x4 gv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * tp + u;
        const f32x4 x = logit_tile<NKH>(img + fa.row + TB * t, eh, el);   // rows = T rows, column = this lane's v
        const f32x4 ls = *reinterpret_cast<const f32x4*>(&sLse[16 * t + 4 * g]);
        const i32x4 yy = *reinterpret_cast<const i32x4*>(&sY[16 * t + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gg = ex2(x[r] + (bv - ls[r])) - (yy[r] == v ? 1.0f : 0.0f);
          gv[u][r] = gg;
          dbsum += gg;
        }
      }
      bf16x8 gh, gl;
      split8(cat(gv[0], gv[1]), gh, gl);
      feed_pair<NKH>(img, fa, 2 * tp, gh, gl, acc);             // dE^T[k][v] += T^T[k][row pair] . g[row pair][v]
    }
    if (cn < m_end) row_finish();                                // the next chunk's lse / labels from the partials requested above
  }
  dbsum += __shfl_xor(dbsum, 16, 64);
  dbsum += __shfl_xor(dbsum, 32, 64);
  if (vlive) {
    float* dst = p.slab + ((int64_t)blockIdx.y * p.V + v) * H;
#pragma unroll
    for (int kb = 0; kb < 2 * NKH; ++kb) *reinterpret_cast<f32x4*>(dst + 16 * kb + 4 * g) = acc[kb];
    if (g == 0) p.bslab[(int64_t)blockIdx.y * p.V + v] = dbsum;
  }
}

// tiles per slice: even, and a whole number of LDS chunks when that costs at most one chunk (a slice of 24 tiles would run
// chunks of 10, 10 and 4 tiles: three staging rounds for 2.4 chunks of work)
int even_tiles(int rows, int slices, int chunk = 0) {
  const int tiles = b4r_cdiv(rows, 16);
  int per = b4r_cdiv(tiles, slices < 1 ? 1 : slices);
  per = (per + 1) & ~1;
  if (per < 2) per = 2;
  if (chunk > 0 && per > chunk) per = b4r_cdiv(per, chunk) * chunk;
  return per;
}

int fwd_slices_wanted(int M) {
  static const int target = getenv("B4R_HEAD_FWD_WGS") ? atoi(getenv("B4R_HEAD_FWD_WGS")) : 480;
  int s = b4r_cdiv(target, b4r_cdiv(M, ROWS_WG));
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}
int dE_slices_wanted(int V) {
  static const int target = getenv("B4R_HEAD_DE_WGS") ? atoi(getenv("B4R_HEAD_DE_WGS")) : 512;
  int s = b4r_cdiv(target, b4r_cdiv(V, ROWS_WG));
  return s < 1 ? 1 : (s > 32 ? 32 : s);
}

constexpr size_t head_lds(int nkh) { return (size_t)head_ch(nkh) * 2 * nkh * IMG_BYTES + 2 * head_ch(nkh) * 16 * sizeof(float); }

template <int NKH>
int launch_fwd(const HeadP& p, int slices, float* dT, float* row_out, float* lse, int32_t* ylab, int only_sweep, hipStream_t stream) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute((const void*)head_fwd_kernel<NKH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)head_lds(NKH));
    raised = true;
  }
  hipLaunchKernelGGL(head_fwd_kernel<NKH>, dim3(b4r_cdiv(p.M, ROWS_WG), slices), dim3(64 * WAVES), head_lds(NKH), stream, p);
  B4R_CHECK_LAUNCH("masked-LM head forward (fused)");
  if (only_sweep) return B4R_OK;
  hipLaunchKernelGGL(head_combine_kernel<NKH>, dim3(b4r_cdiv((int64_t)p.M * 8 * NKH, 256)), dim3(256), 0, stream,
                     (const float*)p.part, slices, p.M, p.V, p.T, p.E, p.bias, p.y, dT, row_out, lse, ylab);
  B4R_CHECK_LAUNCH("masked-LM head combine");
  return B4R_OK;
}

template <int NKH, bool FOLD = false>
int launch_dE(const HeadP& p, int slices, hipStream_t stream) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute((const void*)head_dE_kernel<NKH, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)head_lds(NKH));
    raised = true;
  }
  hipLaunchKernelGGL((head_dE_kernel<NKH, FOLD>), dim3(b4r_cdiv(p.V, ROWS_WG), slices), dim3(64 * WAVES), head_lds(NKH), stream, p);
  B4R_CHECK_LAUNCH("masked-LM head dE (fused)");
  return B4R_OK;
}

}  // namespace

// b4r_head32.hip: the 32 x 32-tile kernels (round 4) that serve hidden size 64 / 128 / 256 unless B4R_HEAD32=0 (B4R_HEAD32_WIDE=0: 64 only)
bool b4r_head32_active(int H);
int b4r_head32_fwd_slices(int M, int V, int H);
int b4r_head32_dE_slices(int M, int V, int H);
int64_t b4r_head32_fwd_scratch_floats(int M, int V, int H);
int64_t b4r_head32_dE_scratch_floats(int M, int V, int H);
int b4r_head32_fwd_launch(const float* T, const float* E, const float* bias, int M, int V, int H, float* scratch, hipStream_t stream);
int b4r_head32_dE_launch(const float* T, const float* E, const float* bias, const float* lse, const int32_t* ylab, int M, int V, int H,
                         float* scratch, hipStream_t stream, const float* fwd_part, int fwd_slices, const int64_t* y, int records_ready);
int b4r_head32_dE_pack_job(const float* T, const float* lse, const int32_t* ylab, int M, int V, int H, float* scratch, const float* fwd_part,
                           int fwd_slices, const int64_t* y, void* out, size_t out_bytes, int* blocks);

bool b4r_head_rx_hidden_ok(int H) { return H == 64 || H == 128 || H == 256; }
int b4r_head_rx_fwd_slices(int M, int V, int H);
// may the dE launch merge the forward's V slices itself (b4r_head_rx_dE_launch with fwd_part)?
bool b4r_head_rx_combine_foldable(int M, int V, int H) {
  static const bool off = getenv("B4R_HEAD_FOLD_COMBINE") && atoi(getenv("B4R_HEAD_FOLD_COMBINE")) == 0;
  return !off && H == 64 && b4r_head_rx_fwd_slices(M, V, H) <= CMAX;
}

// number of V slices the forward uses / M slices the dE kernel uses, and the scratch they need (floats)
int b4r_head_rx_fwd_slices(int M, int V, int H) {
  if (b4r_head32_active(H)) return b4r_head32_fwd_slices(M, V, H);
  const int per = even_tiles(V, fwd_slices_wanted(M), head_ch(H / 32));
  return b4r_cdiv(b4r_cdiv(V, 16), per);
}
int64_t b4r_head_rx_fwd_scratch_floats(int M, int V, int H) {
  if (b4r_head32_active(H)) return b4r_head32_fwd_scratch_floats(M, V, H);
  return (int64_t)b4r_head_rx_fwd_slices(M, V, H) * M * (H + 8 + 2);
}
int b4r_head_rx_dE_slices(int M, int V, int H) {
  if (b4r_head32_active(H)) return b4r_head32_dE_slices(M, V, H);
  const int per = even_tiles(M, dE_slices_wanted(V), head_ch(H / 32));
  return b4r_cdiv(b4r_cdiv(M, 16), per);
}
int64_t b4r_head_rx_dE_scratch_floats(int M, int V, int H) {
  if (b4r_head32_active(H)) return b4r_head32_dE_scratch_floats(M, V, H);
  return (int64_t)b4r_head_rx_dE_slices(M, V, H) * ((int64_t)V * H + V);
}

int b4r_head_rx_fwd_launch2(const float* T, const float* E, const float* bias, const int64_t* y, int M, int V, int H,
                            float* scratch, float* dT, float* row_out, float* lse, int32_t* ylab, int only_sweep,
                            hipStream_t stream) {
  if (b4r_head32_active(H)) {   // the sweep on 32 x 32 tiles, the merge launch (if any) as before
    const int rc = b4r_head32_fwd_launch(T, E, bias, M, V, H, scratch, stream);
    if (rc || only_sweep) return rc;
    const int sl = b4r_head32_fwd_slices(M, V, H);
    const dim3 grid(b4r_cdiv((int64_t)M * (H / 4), 256));
    if (H == 64) hipLaunchKernelGGL(head_combine_kernel<2>, grid, dim3(256), 0, stream, (const float*)scratch, sl, M, V, T, E, bias, y, dT, row_out, lse, ylab);
    else if (H == 128) hipLaunchKernelGGL(head_combine_kernel<4>, grid, dim3(256), 0, stream, (const float*)scratch, sl, M, V, T, E, bias, y, dT, row_out, lse, ylab);
    else hipLaunchKernelGGL(head_combine_kernel<8>, grid, dim3(256), 0, stream, (const float*)scratch, sl, M, V, T, E, bias, y, dT, row_out, lse, ylab);
    B4R_CHECK_LAUNCH("masked-LM head combine");
    return B4R_OK;
  }
  HeadP p{};
  p.T = T; p.E = E; p.bias = bias; p.y = y; p.part = scratch; p.M = M; p.V = V;
  const int slices = b4r_head_rx_fwd_slices(M, V, H);
  p.tiles_per_slice = even_tiles(V, fwd_slices_wanted(M), head_ch(H / 32));
  switch (H) {
    case 64: return launch_fwd<2>(p, slices, dT, row_out, lse, ylab, only_sweep, stream);
    case 128: return launch_fwd<4>(p, slices, dT, row_out, lse, ylab, only_sweep, stream);
    case 256: return launch_fwd<8>(p, slices, dT, row_out, lse, ylab, only_sweep, stream);
    default: b4r_set_error("fused masked-LM head: hidden size %d not supported (64, 128, 256)", H); return B4R_E_SHAPE;
  }
}

// loss rows (as b4r_softmax_ce writes them), lse, labels and dT from T, E, bias, y
int b4r_head_rx_fwd_launch(const float* T, const float* E, const float* bias, const int64_t* y, int M, int V, int H,
                           float* scratch, float* dT, float* row_out, float* lse, int32_t* ylab, hipStream_t stream) {
  return b4r_head_rx_fwd_launch2(T, E, bias, y, M, V, H, scratch, dT, row_out, lse, ylab, 0, stream);
}

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);

// dE [V,H] and db [V] (overwritten, through the ordered slab reduction) from T, E, bias and the forward's lse / labels
// fwd_part != NULL: the forward ran with only_sweep = 1 and left its per-slice partials there; lse / ylab are not read (y: the labels),
// the merge itself (dT, loss rows) is the business of the LayerNorm backward behind this launch (b4r_ln_bwd_launch's merge argument)
// the rider job that forms dE's tile records (32 x 32-tile kernels only: returns 0 blocks otherwise); a following b4r_head_rx_dE_launch
// with records_ready = 1 then skips its own conversion launch
int b4r_head_rx_dE_pack_job(const float* T, const float* lse, const int32_t* ylab, int M, int V, int H, float* scratch,
                            const float* fwd_part, const int64_t* y, void* out, size_t out_bytes, int* blocks) {
  *blocks = 0;
  if (!b4r_head32_active(H)) return B4R_OK;
  return b4r_head32_dE_pack_job(T, lse, ylab, M, V, H, scratch, fwd_part, b4r_head32_fwd_slices(M, V, H), y, out, out_bytes, blocks);
}
int b4r_head_rx_dE_launch(const float* T, const float* E, const float* bias, const float* lse, const int32_t* ylab, int M, int V,
                          int H, float* scratch, float* dE, float* db, hipStream_t stream, const float* fwd_part, const int64_t* y,
                          int records_ready) {
  if (b4r_head32_active(H)) {
    if (fwd_part != nullptr)
      B4R_CHECK_ARG(b4r_head_rx_combine_foldable(M, V, H), B4R_E_BADARG, "fused masked-LM head: the merge cannot ride on dE for this shape");
    const int rc = b4r_head32_dE_launch(T, E, bias, lse, ylab, M, V, H, scratch, stream, fwd_part, b4r_head32_fwd_slices(M, V, H), y, records_ready);
    if (rc) return rc;
    const int sl = b4r_head32_dE_slices(M, V, H);
    return b4r_launch_slab_reduce_full(scratch, sl, V, H, dE, H, 0, nullptr, nullptr, scratch + (int64_t)sl * V * H, db, stream);
  }
  HeadP p{};
  p.T = T; p.E = E; p.bias = bias; p.lse = lse; p.ylab = ylab; p.M = M; p.V = V;
  if (fwd_part != nullptr) {
    B4R_CHECK_ARG(b4r_head_rx_combine_foldable(M, V, H), B4R_E_BADARG, "fused masked-LM head: the merge cannot ride on dE for this shape");
    p.cpart = fwd_part; p.cslices = b4r_head_rx_fwd_slices(M, V, H); p.y = y;
  }
  const int slices = b4r_head_rx_dE_slices(M, V, H);
  p.tiles_per_slice = even_tiles(M, dE_slices_wanted(V), head_ch(H / 32));
  p.slab = scratch;
  p.bslab = scratch + (int64_t)slices * V * H;
  int rc;
  switch (H) {
    case 64: rc = fwd_part ? launch_dE<2, true>(p, slices, stream) : launch_dE<2>(p, slices, stream); break;
    case 128: rc = launch_dE<4>(p, slices, stream); break;
    case 256: rc = launch_dE<8>(p, slices, stream); break;
    default: b4r_set_error("fused masked-LM head: hidden size %d not supported (64, 128, 256)", H); return B4R_E_SHAPE;
  }
  if (rc) return rc;
  return b4r_launch_slab_reduce_full(p.slab, slices, V, H, dE, H, 0, nullptr, nullptr, p.bslab, db, stream);
}

extern "C" int64_t b4r_mlm_head_fused_scratch_floats(int32_t M, int32_t V, int32_t H) {
  if (M <= 0 || V <= 0 || !b4r_head_rx_hidden_ok(H)) return 0;
  const int64_t a = b4r_head_rx_fwd_scratch_floats(M, V, H), b = b4r_head_rx_dE_scratch_floats(M, V, H);
  return a > b ? a : b;
}

extern "C" int b4r_mlm_head_fused_fwd(const float* T, const float* E, const float* bias, const int64_t* y_true, int32_t M,
                                      int32_t V, int32_t H, float* scratch, float* dT, float* row_scratch, float* lse,
                                      int32_t* labels, int32_t only_sweep, b4r_stream_t stream) {
  B4R_CHECK_ARG(T && E && bias && y_true && scratch, B4R_E_BADARG, "b4r_mlm_head_fused_fwd: null argument");
  B4R_CHECK_ARG(only_sweep || (dT && row_scratch && lse && labels), B4R_E_BADARG, "b4r_mlm_head_fused_fwd: null output");
  B4R_CHECK_ARG(M > 0 && V > 0 && b4r_head_rx_hidden_ok(H), B4R_E_SHAPE, "b4r_mlm_head_fused_fwd: bad shape (H = 64, 128 or 256)");
  B4R_CHECK_ARG(b4r_aligned16(T) && b4r_aligned16(E) && b4r_aligned16(scratch) && (only_sweep || b4r_aligned16(dT)), B4R_E_ALIGN,
                "b4r_mlm_head_fused_fwd: T, E, scratch and dT must be 16-byte aligned");
  return b4r_head_rx_fwd_launch2(T, E, bias, y_true, M, V, H, scratch, dT, row_scratch, lse, labels, only_sweep, (hipStream_t)stream);
}

extern "C" int b4r_mlm_head_fused_bwd(const float* T, const float* E, const float* bias, const float* lse, const int32_t* labels,
                                     int32_t M, int32_t V, int32_t H, float* scratch, float* dE, float* dbias,
                                     b4r_stream_t stream) {
  B4R_CHECK_ARG(T && E && bias && lse && labels && scratch && dE && dbias, B4R_E_BADARG, "b4r_mlm_head_fused_bwd: null argument");
  B4R_CHECK_ARG(M > 0 && V > 0 && b4r_head_rx_hidden_ok(H), B4R_E_SHAPE, "b4r_mlm_head_fused_bwd: bad shape (H = 64, 128 or 256)");
  B4R_CHECK_ARG(b4r_aligned16(T) && b4r_aligned16(E) && b4r_aligned16(scratch), B4R_E_ALIGN,
                "b4r_mlm_head_fused_bwd: T, E and scratch must be 16-byte aligned");
  return b4r_head_rx_dE_launch(T, E, bias, lse, labels, M, V, H, scratch, dE, dbias, (hipStream_t)stream, nullptr, nullptr, 0);
}
