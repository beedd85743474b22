// Dense layers on the exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain).
//
//   gemm_kernel     C[M,N] = A[M,K] . B      B as [K,N] (forward: Keras kernels are [in,out]) or as [N,K]
//                   (C = A.B^T: input gradients, and the tied vocabulary projection T.E^T of tfm MaskedLM)
//                   128x64 output tile per 256-thread workgroup, K staged 32 at a time through LDS, fused epilogues.
//   gemm_tn_kernel  out[Mo,No] = A[R,Mo]^T . B[R,No]   weight gradients: rows R split over workgroups, partial
//                   slabs + ordered reduce (bitwise reproducible, no float atomics), bias column sums fused.
//
// MFMA 32x32x2 f32 operand map (guide §3): lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
// The k index fed to the two lane halves is permuted (half h takes k = 8t+4h+r) so that one ds_read_b128 serves four
// MFMA steps; A and B use the same permutation, so the sum over k is unchanged.
#include "b4r_common.h"

namespace {

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int LDS_A = BK + 4;   // 36 floats: conflict-free ds_read_b128 down a column of 16 rows
constexpr int LDS_BN = BN;      // B as [k][n]: ds_read_b32 across n, conflict-free
constexpr int LDS_BK = BK + 4;  // B as [n][k]

struct GemmP {
  const float* A; const float* B; float* C; const float* bias; float* C2; const float* R;
  int lda, ldb, ldc, ldc2, ldr;
  int M, N, K;
  int tiles_n;
  int k_chunk;            // K range per blockIdx.y (multiple of BK); == K rounded up when not split
  int64_t split_stride;   // floats between the C slabs of consecutive K splits
  int a_vec, b_vec;
  float qscale; int qcols;
  DropArgs drop;
};

__device__ __forceinline__ f32x4 load4_guard(const float* base, int64_t off, int n_valid, bool vec) {
  // n_valid: how many of the 4 consecutive elements are in range (<=0: none)
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (n_valid >= 4 && vec) {
    v = *reinterpret_cast<const f32x4*>(base + off);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < n_valid) v[e] = base[off + e];
  }
  return v;
}

template <bool B_NK, int EPI, bool A_DROP>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) float sA[BM * LDS_A];
  __shared__ __attribute__((aligned(16))) float sB[B_NK ? BN * LDS_BK : BK * LDS_BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int h = lane >> 5, l31 = lane & 31;

  DropCtx dctx = b4r_drop_ctx(p.drop);

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  f32x4 ra[4], rb[2];

  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3, c4 = (f & 7) * 4;
      const int gr = m0 + row, gk = k0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gr < p.M) v = load4_guard(p.A, (int64_t)gr * p.lda + gk, p.K - gk, p.a_vec != 0);
      if (A_DROP) {
        if (dctx.on) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = b4r_drop(dctx, v[e], (uint64_t)gr * (uint64_t)p.K + (uint64_t)(gk + e));
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (B_NK) {
        const int nr = f >> 3, c4 = (f & 7) * 4;
        const int gn = n0 + nr, gk = k0 + c4;
        if (gn < p.N) v = load4_guard(p.B, (int64_t)gn * p.ldb + gk, p.K - gk, p.b_vec != 0);
      } else {
        const int kr = f >> 4, c4 = (f & 15) * 4;
        const int gk = k0 + kr, gn = n0 + c4;
        if (gk < p.K) v = load4_guard(p.B, (int64_t)gk * p.ldb + gn, p.N - gn, p.b_vec != 0);
      }
      rb[i] = v;
    }
  };

  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = f >> 3, c4 = (f & 7) * 4;
      *reinterpret_cast<f32x4*>(&sA[row * LDS_A + c4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      if (B_NK) {
        const int nr = f >> 3, c4 = (f & 7) * 4;
        *reinterpret_cast<f32x4*>(&sB[nr * LDS_BK + c4]) = rb[i];
      } else {
        const int kr = f >> 4, c4 = (f & 15) * 4;
        *reinterpret_cast<f32x4*>(&sB[kr * LDS_BN + c4]) = rb[i];
      }
    }
  };

  const int k_begin = blockIdx.y * p.k_chunk;
  const int k_end = min(p.K, k_begin + p.k_chunk);
  const int nk = (k_end - k_begin + BK - 1) / BK;
  if (blockIdx.y > 0) p.C += (int64_t)blockIdx.y * p.split_stride;
  if (nk > 0) load_tiles(k_begin);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (kt + 1 < nk) load_tiles(k_begin + (kt + 1) * BK);
    const int arow = wave * 32 + l31;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&sA[arow * LDS_A + 8 * t + 4 * h]);
      if (B_NK) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(&sB[l31 * LDS_BK + 8 * t + 4 * h]);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(&sB[(32 + l31) * LDS_BK + 8 * t + 4 * h]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b0[r], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b1[r], acc1, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = 8 * t + 4 * h + r;
          const float b0 = sB[kk * LDS_BN + l31];
          const float b1 = sB[kk * LDS_BN + 32 + l31];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b1, acc1, 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------------
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int col = n0 + jt * 32 + l31;
    if (col >= p.N) continue;
    float bv = 0.f;
    if (EPI == B4R_EPI_BIAS || EPI == B4R_EPI_BIAS_QSCALE || EPI == B4R_EPI_BIAS_GELU || EPI == B4R_EPI_BIAS_DROP_RES ||
        EPI == B4R_EPI_BIAS_TANH)
      bv = p.bias[col];
    const float qs = (EPI == B4R_EPI_BIAS_QSCALE && col < p.qcols) ? p.qscale : 1.0f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row >= p.M) continue;
      float v = (jt == 0 ? acc0[reg] : acc1[reg]);
      const int64_t co = (int64_t)row * p.ldc + col;
      if (EPI == B4R_EPI_NONE) {
        p.C[co] = v;
      } else if (EPI == B4R_EPI_BIAS) {
        p.C[co] = v + bv;
      } else if (EPI == B4R_EPI_BIAS_QSCALE) {
        p.C[co] = (v + bv) * qs;
      } else if (EPI == B4R_EPI_BIAS_GELU) {
        const float pre = v + bv;
        p.C2[(int64_t)row * p.ldc2 + col] = pre;
        p.C[co] = b4r_gelu(pre);
      } else if (EPI == B4R_EPI_BIAS_DROP_RES) {
        float y = v + bv;
        y = b4r_drop(dctx, y, (uint64_t)row * (uint64_t)p.N + (uint64_t)col);
        p.C[co] = p.R[(int64_t)row * p.ldr + col] + y;
      } else if (EPI == B4R_EPI_GELU_BWD) {
        p.C[co] = v * b4r_gelu_grad(p.R[(int64_t)row * p.ldr + col]);
      } else if (EPI == B4R_EPI_ADD_RES) {
        p.C[co] = v + p.R[(int64_t)row * p.ldr + col];
      } else if (EPI == B4R_EPI_BIAS_TANH) {
        p.C[co] = tanhf(v + bv);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// TN: out[Mo,No] = sum_r A[r,Mo]^T B[r,No]
// ---------------------------------------------------------------------------------------------------------------
constexpr int TB = 64;   // output tile 64 x 64
constexpr int TK = 32;   // rows per stage

struct TnP {
  const float* A; const float* B; float* slab; float* colsum_slab; float* colsum_a_slab;
  int lda, ldb;
  int R, Mo, No;
  int chunk;  // rows per z-slice (multiple of TK)
  int a_vec, b_vec;
  DropArgs drop;
};

template <bool B_DROP>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnP p) {
  __shared__ __attribute__((aligned(16))) float sA[TK * TB];
  __shared__ __attribute__((aligned(16))) float sB[TK * TB];
  __shared__ float sRed[4 * TB];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.x * TB, j0 = blockIdx.y * TB, z = blockIdx.z;
  const int r_begin = z * p.chunk;
  const int r_end = min(p.R, r_begin + p.chunk);
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const bool do_colsum = (p.colsum_slab != nullptr) && (blockIdx.x == 0);
  const bool do_colsum_a = (p.colsum_a_slab != nullptr) && (blockIdx.y == 0);

  DropCtx dctx = b4r_drop_ctx(p.drop);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float cs = 0.f, csa = 0.f;

  f32x4 ra[2], rb[2];
  auto load_tiles = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      const int rr = f >> 4, c4 = (f & 15) * 4;
      const int gr = r0 + rr;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (gr < r_end) {
        va = load4_guard(p.A, (int64_t)gr * p.lda + i0 + c4, p.Mo - (i0 + c4), p.a_vec != 0);
        vb = load4_guard(p.B, (int64_t)gr * p.ldb + j0 + c4, p.No - (j0 + c4), p.b_vec != 0);
        if (B_DROP) {
          if (dctx.on) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              vb[e] = b4r_drop(dctx, vb[e], (uint64_t)gr * (uint64_t)p.No + (uint64_t)(j0 + c4 + e));
          }
        }
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      const int rr = f >> 4, c4 = (f & 15) * 4;
      *reinterpret_cast<f32x4*>(&sA[rr * TB + c4]) = ra[i];
      *reinterpret_cast<f32x4*>(&sB[rr * TB + c4]) = rb[i];
    }
  };

  if (r_begin < r_end) load_tiles(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += TK) {
    __syncthreads();
    store_tiles();
    __syncthreads();
    if (r0 + TK < r_end) load_tiles(r0 + TK);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float a = sA[(2 * s + h) * TB + 32 * wm + l31];
      const float b = sB[(2 * s + h) * TB + 32 * wn + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (do_colsum) {
#pragma unroll
      for (int r = 0; r < 8; ++r) cs += sB[(wave * 8 + r) * TB + lane];
    }
    if (do_colsum_a) {
#pragma unroll
      for (int r = 0; r < 8; ++r) csa += sA[(wave * 8 + r) * TB + lane];
    }
  }

  const int col = j0 + 32 * wn + l31;
  if (col < p.No) {
    float* slab = p.slab + (int64_t)z * p.Mo * p.No;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = i0 + 32 * wm + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (row < p.Mo) slab[(int64_t)row * p.No + col] = acc[reg];
    }
  }
  if (do_colsum) {
    sRed[wave * TB + lane] = cs;
    __syncthreads();
    if (tid < TB && j0 + tid < p.No)
      p.colsum_slab[(int64_t)z * p.No + j0 + tid] = (sRed[tid] + sRed[TB + tid]) + (sRed[2 * TB + tid] + sRed[3 * TB + tid]);
  }
  if (do_colsum_a) {
    __syncthreads();
    sRed[wave * TB + lane] = csa;
    __syncthreads();
    if (tid < TB && i0 + tid < p.Mo)
      p.colsum_a_slab[(int64_t)z * p.Mo + i0 + tid] = (sRed[tid] + sRed[TB + tid]) + (sRed[2 * TB + tid] + sRed[3 * TB + tid]);
  }
}

// out[row*ldo+col] (+)= sum_z slab[z][row][col].  A workgroup owns 256 consecutive elements of the matrix (a lane = 4 of
// them, one 16-byte load per slab) or 256 elements of a column-sum strip (lanes 0..63 of each wave, scalar loads); its 16
// waves each sum every 16th slab with independent loads in flight, then the 16 partial sums are combined in a fixed
// order: bitwise reproducible.  16-byte loads need (Mo * No) % 4 == 0 and 16-byte aligned slabs (the model's scratch
// regions are); otherwise the same lanes fall back to 4 scalar loads per slab (same summation order).
constexpr int RZ = 16;
constexpr int RE = 256;    // elements per workgroup when the slabs are shared out over the waves
constexpr int RQ = 4096;   // elements per workgroup when every thread walks all (<= 16) slabs itself
typedef B4rReduceJob ReduceJobView;
__device__ __forceinline__ int reduce_mat_blocks(int S, int64_t total) { return (int)((total + (S <= RZ ? RQ : RE) - 1) / (S <= RZ ? RQ : RE)); }
__device__ __forceinline__ float reduce_store(const ReduceJobView& job, int64_t eo, float r) {
  const int row = (int)(eo / job.No), col = (int)(eo % job.No);
  float* o = job.out + (int64_t)row * job.ldo + col;
  if (job.fix != nullptr) {   // integer sum first (order-free), one conversion, one float add
    long long q = job.fix[eo];
    if (eo < job.fix_hot_elems) {   // 16 loads in flight: one after the other the 64 slots were 30 us of latency in one workgroup
      for (int s0 = 0; s0 < job.fix_slots; s0 += 16) {
        long long v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s0 + u < job.fix_slots ? job.fix_hot[(int64_t)(s0 + u) * job.fix_hot_elems + eo] : 0ll;
#pragma unroll
        for (int u = 0; u < 16; ++u) q += v[u];
      }
    }
    r += (float)q * (1.f / 68719476736.f);   // units of 2^-36 (b4r_rowops.hip: FIX_SCALE)
    if (job.fix_poison != nullptr && *job.fix_poison != 0) r = __builtin_nanf("");   // the scatter met Inf / NaN / |x| >= 2^18
  }
  if (job.accumulate) r += *o;
  *o = r;
  return r;
}
// returns the sum of the squares of the values THIS thread stored (the global gradient norm's partial sums, multi_slab_reduce_kernel)
__device__ __forceinline__ float slab_reduce_block(const ReduceJobView& job, int block, float (*sp)[RE]) {
  float sq = 0.f;
  const int64_t total = (int64_t)job.Mo * job.No;
  const int mat_blocks = reduce_mat_blocks(job.S, total);
  const int lane = threadIdx.x & 63, zl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool vec = (total % 4 == 0) && ((reinterpret_cast<uintptr_t>(job.slab) & 15) == 0);   // block-uniform
  if (block < mat_blocks && job.S <= RZ) {
    // few slabs (the head's table gradient: 16 slabs of V x H): a thread owns 4 elements and reads all slabs itself, every load in
    // flight at once.  Same value as the shared-out form below: there wave z's partial sum is slab z alone.
    const int64_t e = (int64_t)block * RQ + 4 * (int64_t)threadIdx.x;
    if (e >= total) return sq;
    f32x4 v[RZ];
#pragma unroll
    for (int z = 0; z < RZ; ++z) {
      v[z] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (z < job.S) {
        if (vec) v[z] = *reinterpret_cast<const f32x4*>(job.slab + (int64_t)z * total + e);
        else
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (e + k < total) v[z][k] = job.slab[(int64_t)z * total + e + k];
      }
    }
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int z = 0; z < RZ; ++z) r += v[z];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (e + k < total) { const float w = reduce_store(job, e + k, r[k]); sq = fmaf(w, w, sq); }
    return sq;
  }
  if (block < mat_blocks) {
    const int64_t e = (int64_t)block * RE + 4 * lane;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      if (e < total) {
        const float* src = job.slab + e;
        // up to 16 slabs per wave (S <= 256): all loads of a wave in flight at once (slabs past the end read as zero), summed
        // in slab order
        for (int z = zl; z < job.S; z += 8 * RZ) {
          f32x4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            v[u] = z + u * RZ < job.S ? *reinterpret_cast<const f32x4*>(src + (int64_t)(z + u * RZ) * total) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < 8; ++u) s += v[u];
        }
      }
    } else {
      for (int z = zl; z < job.S; z += RZ) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (e + c < total) s[c] += job.slab[(int64_t)z * total + e + c];
      }
    }
    *reinterpret_cast<f32x4*>(&sp[zl][4 * lane]) = s;
    __syncthreads();
    if (zl < 4) {                       // 4 waves x 64 lanes finish one element each
      const int l = threadIdx.x;        // 0..255
      const int64_t eo = (int64_t)block * RE + l;
      if (eo < total) {
        float r = 0.f;
#pragma unroll
        for (int z = 0; z < RZ; ++z) r += sp[z][l];
        const float v = reduce_store(job, eo, r);
        sq = v * v;
      }
    }
    return sq;
  }
  // column-sum strips: [No of cslab | Mo of caslab], 64 elements per wave-row, 4 wave-rows of the 16 x 64 layout unused
  const int64_t n_cs = job.colsum ? job.No : 0, n_csa = job.colsum_a ? job.Mo : 0;
  const int64_t e = (int64_t)(block - mat_blocks) * 64 + lane;
  const float* src = nullptr;
  int64_t stride = 0, idx = 0;
  if (e < n_cs) { src = job.cslab; stride = job.No; idx = e; }
  else if (e < n_cs + n_csa) { src = job.caslab; stride = job.Mo; idx = e - n_cs; }
  float s = 0.f;
  if (src) {
    for (int z = zl; z < job.S; z += 8 * RZ) {   // 8 loads in flight, summed in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = z + u * RZ < job.S ? src[(int64_t)(z + u * RZ) * stride + idx] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  }
  sp[zl][lane] = s;
  __syncthreads();
  if (zl == 0 && src) {
    float r = 0.f;
#pragma unroll
    for (int z = 0; z < RZ; ++z) r += sp[z][lane];
    if (e < n_cs) job.colsum[idx] = r;
    else job.colsum_a[idx] = r;
    sq = r * r;
  }
  return sq;
}

__global__ __launch_bounds__(64 * RZ) void slab_reduce_kernel(ReduceJobView job) {
  __shared__ __attribute__((aligned(16))) float sp[RZ][RE];
  (void)slab_reduce_block(job, (int)blockIdx.x, sp);
}

int slab_reduce_grid(int S, int Mo, int No, bool cs, bool csa) {
  return b4r_cdiv((int64_t)Mo * No, S <= RZ ? RQ : RE) + b4r_cdiv((int64_t)(cs ? No : 0) + (csa ? Mo : 0), 64);
}

int tn_split(int R, int Mo, int No) {
  // about four workgroups per CU in total (each keeps only one 16 KB stage in flight, so residency hides the HBM
  // latency); at most 256 slabs so that the ordered reduction stays short
  const int tiles = b4r_cdiv(Mo, TB) * b4r_cdiv(No, TB);
  int S = b4r_cdiv(1024, tiles);
  const int max_s = b4r_cdiv(R, 4 * TK);  // at least 128 rows per slice
  if (S > max_s) S = max_s;
  if (S > 256) S = 256;
  if (S < 1) S = 1;
  return S;
}

template <bool B_NK, int EPI>
void launch_gemm(const GemmP& p, int a_drop, dim3 grid, hipStream_t s) {
  if (a_drop)
    hipLaunchKernelGGL((gemm_kernel<B_NK, EPI, true>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((gemm_kernel<B_NK, EPI, false>), grid, dim3(256), 0, s, p);
}

template <bool B_NK>
int dispatch_epi(const GemmP& p, int epi, int a_drop, dim3 grid, hipStream_t s) {
  switch (epi) {
    case B4R_EPI_NONE: launch_gemm<B_NK, B4R_EPI_NONE>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS: launch_gemm<B_NK, B4R_EPI_BIAS>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_QSCALE: launch_gemm<B_NK, B4R_EPI_BIAS_QSCALE>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_GELU: launch_gemm<B_NK, B4R_EPI_BIAS_GELU>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_DROP_RES: launch_gemm<B_NK, B4R_EPI_BIAS_DROP_RES>(p, a_drop, grid, s); break;
    case B4R_EPI_GELU_BWD: launch_gemm<B_NK, B4R_EPI_GELU_BWD>(p, a_drop, grid, s); break;
    case B4R_EPI_ADD_RES: launch_gemm<B_NK, B4R_EPI_ADD_RES>(p, a_drop, grid, s); break;
    case B4R_EPI_BIAS_TANH: launch_gemm<B_NK, B4R_EPI_BIAS_TANH>(p, a_drop, grid, s); break;
    default: b4r_set_error("b4r_gemm_f32: unknown epilogue %d", epi); return B4R_E_BADARG;
  }
  return B4R_OK;
}

}  // namespace

int b4r_launch_slab_reduce(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                           hipStream_t stream) {
  ReduceJobView job{slab, nullptr, nullptr, out, nullptr, nullptr, S, Mo, No, ldo, accumulate};
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(slab_reduce_grid(S, Mo, No, false, false)), dim3(64 * RZ), 0, stream, job);
  B4R_CHECK_LAUNCH("slab_reduce");
  return B4R_OK;
}

namespace {
struct MultiReduceP {
  B4rReduceJob jobs[B4R_MAX_REDUCE_JOBS];
  int block_begin[B4R_MAX_REDUCE_JOBS + 1];
  int n;
  float* sq_partial;   // optional [gridDim.x]: sum of the squares of everything the workgroup stored
};
// all queued reductions in one launch: a workgroup looks up its job, then does what slab_reduce_kernel does
__global__ __launch_bounds__(64 * RZ) void multi_slab_reduce_kernel(MultiReduceP p) {
  __shared__ __attribute__((aligned(16))) float sp[RZ][RE];
  int j = 0;
  while (j + 1 < p.n && (int)blockIdx.x >= p.block_begin[j + 1]) ++j;
  const B4rReduceJob q = p.jobs[j];
  float sq = slab_reduce_block(q, (int)blockIdx.x - p.block_begin[j], sp);
  if (p.sq_partial != nullptr) {   // fixed order: wave butterfly, then the 16 waves in turn
    __shared__ float s_sq[RZ];
    sq = b4r_wave_sum(sq);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int z = 0; z < RZ; ++z) t += s_sq[z];
      p.sq_partial[blockIdx.x] = t;
    }
  }
}
thread_local B4rReduceQueue* g_queue = nullptr;
}  // namespace

void b4r_reduce_queue_begin(B4rReduceQueue* q) { q->n = 0; g_queue = q; }
bool b4r_reduce_queue_push(const B4rReduceJob& job) {
  if (g_queue == nullptr || g_queue->n >= B4R_MAX_REDUCE_JOBS) return false;
  g_queue->jobs[g_queue->n++] = job;
  return true;
}
bool b4r_reduce_queue_attach_fixed(const float* out, const long long* fix, const long long* fix_hot, int fix_hot_elems, int fix_slots,
                                   const int* fix_poison) {
  if (g_queue == nullptr) return false;
  for (int j = 0; j < g_queue->n; ++j) {
    B4rReduceJob& job = g_queue->jobs[j];
    if (job.out == out && job.ldo == job.No && job.fix == nullptr) {
      job.fix = fix; job.fix_hot = fix_hot; job.fix_hot_elems = fix_hot_elems; job.fix_slots = fix_slots; job.fix_poison = fix_poison;
      return true;
    }
  }
  return false;
}
int b4r_launch_reduce_job(const B4rReduceJob& job, hipStream_t stream) {
  if (b4r_reduce_queue_push(job)) return B4R_OK;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(slab_reduce_grid(job.S, job.Mo, job.No, job.colsum != nullptr, job.colsum_a != nullptr)),
                     dim3(64 * RZ), 0, stream, job);
  B4R_CHECK_LAUNCH("slab_reduce");
  return B4R_OK;
}
int b4r_reduce_queue_flush(hipStream_t stream, float* sq_partial, int sq_cap, int* sq_np, int64_t* covered) {
  B4rReduceQueue* q = g_queue;
  g_queue = nullptr;
  if (sq_np) *sq_np = 0;
  if (covered) *covered = 0;
  if (q == nullptr || q->n == 0) return B4R_OK;
  MultiReduceP p;
  p.n = q->n;
  int blocks = 0;
  int64_t elems = 0;
  for (int j = 0; j < q->n; ++j) {
    p.jobs[j] = q->jobs[j];
    p.block_begin[j] = blocks;
    blocks += slab_reduce_grid(q->jobs[j].S, q->jobs[j].Mo, q->jobs[j].No, q->jobs[j].colsum != nullptr, q->jobs[j].colsum_a != nullptr);
    elems += (int64_t)q->jobs[j].Mo * q->jobs[j].No + (q->jobs[j].colsum ? q->jobs[j].No : 0) + (q->jobs[j].colsum_a ? q->jobs[j].Mo : 0);
  }
  p.block_begin[q->n] = blocks;
  // the partial square sums only when they fit the caller's array (else the caller measures the norm with a launch of its own)
  p.sq_partial = (sq_partial != nullptr && blocks <= sq_cap) ? sq_partial : nullptr;
  if (p.sq_partial && sq_np) *sq_np = blocks;
  if (covered) *covered = elems;
  hipLaunchKernelGGL(multi_slab_reduce_kernel, dim3(blocks), dim3(64 * RZ), 0, stream, p);
  B4R_CHECK_LAUNCH("multi_slab_reduce");
  return B4R_OK;
}

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream) {
  B4rReduceJob job{slab, cslab, caslab, out, colsum, colsum_a, S, Mo, No, ldo, accumulate};
  if (b4r_reduce_queue_push(job)) return B4R_OK;
  ReduceJobView view{slab, cslab, caslab, out, colsum, colsum_a, S, Mo, No, ldo, accumulate};
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(slab_reduce_grid(S, Mo, No, colsum != nullptr, colsum_a != nullptr)), dim3(64 * RZ),
                     0, stream, view);
  B4R_CHECK_LAUNCH("slab_reduce");
  return B4R_OK;
}

// ---- arithmetic mode of the dense layers ------------------------------------------------------------------------------
int b4r_gemm_rx_launch(const b4r_gemm_desc* d, hipStream_t stream);
bool b4r_gemm_rx_supported(const b4r_gemm_desc* d);
bool b4r_gemm_rx_tn_supported(const b4r_gemm_tn_desc* d);
int64_t b4r_gemm_rx_tn_scratch_floats(int R, int Mo, int No);
int b4r_gemm_rx_tn_launch(const b4r_gemm_tn_desc* d, float* scratch, hipStream_t stream);
bool b4r_gemm_rx_tn_pair_supported(const b4r_gemm_tn_desc* d0, const b4r_gemm_tn_desc* d1);
int b4r_gemm_rx_tn_pair_launch(const b4r_gemm_tn_desc* d0, float* scratch0, const b4r_gemm_tn_desc* d1, float* scratch1,
                               hipStream_t stream);
static int g_gemm_mode = B4R_GEMM_BF16X3;
extern "C" int b4r_set_gemm_mode(int mode) {
  B4R_CHECK_ARG(mode == B4R_GEMM_F32 || mode == B4R_GEMM_BF16X3, B4R_E_BADARG, "b4r_set_gemm_mode: unknown mode %d", mode);
  g_gemm_mode = mode;
  return B4R_OK;
}
extern "C" int b4r_get_gemm_mode(void) { return g_gemm_mode; }

extern "C" int64_t b4r_gemm_ln_bwd_partial_floats(int32_t M) { return (int64_t)b4r_cdiv(M > 0 ? M : 1, 64) * 128; }
extern "C" int b4r_gemm_ln_supported(const b4r_gemm_desc* d) {
  if (d == nullptr || g_gemm_mode != B4R_GEMM_BF16X3 || !d->A || !d->B || !d->C || d->M <= 0) return 0;
  if (d->epilogue == B4R_EPI_BIAS_GELU_LN) return b4r_gemm_rx_supported(d) ? 1 : 0;
  return ((d->epilogue == B4R_EPI_BIAS_DROP_RES_LN || d->epilogue == B4R_EPI_ADD_RES_LN_BWD) && d->R && d->ldr >= d->N &&
          b4r_gemm_rx_supported(d)) ? 1 : 0;
}

extern "C" int b4r_gemm_f32(const b4r_gemm_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_gemm_f32: null descriptor");
  B4R_CHECK_ARG(d->A && d->B && d->C, B4R_E_BADARG, "b4r_gemm_f32: null operand");
  B4R_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, B4R_E_SHAPE, "b4r_gemm_f32: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  B4R_CHECK_ARG(d->lda >= d->K && d->ldc >= d->N && d->ldb >= (d->b_is_nk ? d->K : d->N), B4R_E_SHAPE,
                "b4r_gemm_f32: leading dimension smaller than the row length");
  const int epi = d->epilogue;
  b4r_timing_detail("M=%d N=%d K=%d epi=%d%s", d->M, d->N, d->K, d->epilogue, d->b_is_nk ? " B^T" : "");
  const bool needs_bias = epi == B4R_EPI_BIAS || epi == B4R_EPI_BIAS_QSCALE || epi == B4R_EPI_BIAS_GELU ||
                          epi == B4R_EPI_BIAS_DROP_RES || epi == B4R_EPI_BIAS_TANH || epi == B4R_EPI_BIAS_DROP_RES_LN ||
                          epi == B4R_EPI_BIAS_GELU_LN;
  B4R_CHECK_ARG(!needs_bias || d->bias, B4R_E_BADARG, "b4r_gemm_f32: epilogue %d needs a bias", epi);
  const bool needs_r = epi == B4R_EPI_BIAS_DROP_RES || epi == B4R_EPI_GELU_BWD || epi == B4R_EPI_ADD_RES ||
                       epi == B4R_EPI_BIAS_DROP_RES_LN || epi == B4R_EPI_ADD_RES_LN_BWD;
  B4R_CHECK_ARG(!needs_r || (d->R && d->ldr >= d->N), B4R_E_BADARG, "b4r_gemm_f32: epilogue %d needs R", epi);
  B4R_CHECK_ARG(epi != B4R_EPI_BIAS_GELU || (d->C2 && d->ldc2 >= d->N), B4R_E_BADARG, "b4r_gemm_f32: BIAS_GELU needs C2");
  B4R_CHECK_ARG(epi >= B4R_EPI_NONE && epi <= B4R_EPI_BIAS_GELU_LN, B4R_E_BADARG, "b4r_gemm_f32: unknown epilogue %d", epi);
  B4R_CHECK_ARG(d->a_gather_idx == nullptr || (epi == B4R_EPI_BIAS_GELU_LN && d->a_gather_add_per > 0 && d->a_gather_per > 0),
                B4R_E_BADARG, "b4r_gemm_f32: a_gather_idx only with B4R_EPI_BIAS_GELU_LN (and a_gather_add_per, a_gather_per > 0)");
  B4R_CHECK_ARG(d->a_copy == nullptr || (d->a_gather_idx != nullptr && d->a_copy_ld >= d->K && d->a_copy_ld % 4 == 0 &&
                                         b4r_aligned16(d->a_copy)),
                B4R_E_BADARG, "b4r_gemm_f32: a_copy needs a_gather_idx, a_copy_ld >= K (multiple of 4) and 16-byte alignment");
  if (epi == B4R_EPI_BIAS_GELU_LN) {
    B4R_CHECK_ARG(d->C2 && d->C3 && d->ln_gamma && d->ln_beta, B4R_E_BADARG, "b4r_gemm_f32: BIAS_GELU_LN needs C2, C3, ln_gamma, ln_beta");
    B4R_CHECK_ARG(b4r_gemm_ln_supported(d), B4R_E_SHAPE,
                  "b4r_gemm_f32: BIAS_GELU_LN not available for M=%d N=%d K=%d in this mode (b4r_gemm_ln_supported)", d->M, d->N,
                  d->K);
  }
  if (epi == B4R_EPI_ADD_RES_LN_BWD) {
    B4R_CHECK_ARG(d->C2 && d->ln_gamma && (d->ln_z || d->ln_ids) && d->ln_mean && d->ln_rstd && d->ln_dgamma, B4R_E_BADARG,
                  "b4r_gemm_f32: ADD_RES_LN_BWD needs C2 (partials), ln_gamma, ln_z (or ln_ids), ln_mean, ln_rstd, ln_dgamma");
    B4R_CHECK_ARG(!d->ln_ids || (d->ln_table && d->ln_pos && d->ln_L > 0 && d->ln_V > 0), B4R_E_BADARG,
                  "b4r_gemm_f32: ADD_RES_LN_BWD with ln_ids needs ln_table, ln_pos, ln_L, ln_V");
    B4R_CHECK_ARG(d->ln_dbeta == d->ln_dgamma + 64, B4R_E_BADARG, "b4r_gemm_f32: ADD_RES_LN_BWD: ln_dbeta must be ln_dgamma + 64");
    B4R_CHECK_ARG(b4r_gemm_ln_supported(d), B4R_E_SHAPE,
                  "b4r_gemm_f32: ADD_RES_LN_BWD not available for M=%d N=%d K=%d in this mode (b4r_gemm_ln_supported)", d->M,
                  d->N, d->K);
    int rc = b4r_gemm_rx_launch(d, (hipStream_t)stream);
    if (rc) return rc;
    return b4r_launch_slab_reduce_full(d->C2, b4r_cdiv(d->M, 64), 1, 128, d->ln_dgamma, 128, 0, nullptr, nullptr, nullptr, nullptr,
                                       (hipStream_t)stream);
  }
  if (epi == B4R_EPI_BIAS_DROP_RES_LN) {
    B4R_CHECK_ARG(d->C2 && d->ln_gamma && d->ln_beta, B4R_E_BADARG, "b4r_gemm_f32: BIAS_DROP_RES_LN needs C2, ln_gamma, ln_beta");
    B4R_CHECK_ARG(b4r_gemm_ln_supported(d), B4R_E_SHAPE,
                  "b4r_gemm_f32: BIAS_DROP_RES_LN not available for M=%d N=%d K=%d in this mode (b4r_gemm_ln_supported)", d->M,
                  d->N, d->K);
  }
  if (g_gemm_mode == B4R_GEMM_BF16X3 && b4r_gemm_rx_supported(d)) return b4r_gemm_rx_launch(d, (hipStream_t)stream);

  GemmP p;
  p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.C2 = d->C2; p.R = d->R;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldc2 = d->ldc2; p.ldr = d->ldr;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.tiles_n = b4r_cdiv(d->N, BN);
  p.k_chunk = b4r_cdiv(d->K, BK) * BK;
  p.split_stride = 0;
  p.a_vec = (b4r_aligned16(d->A) && (d->lda % 4 == 0)) ? 1 : 0;
  p.b_vec = (b4r_aligned16(d->B) && (d->ldb % 4 == 0)) ? 1 : 0;
  p.qscale = d->qscale; p.qcols = d->qcols;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  const int a_drop = (d->a_dropout && p.drop.rng != nullptr) ? 1 : 0;
  const int64_t tiles = (int64_t)b4r_cdiv(d->M, BM) * p.tiles_n;
  B4R_CHECK_ARG(tiles < 2147483647LL, B4R_E_SHAPE, "b4r_gemm_f32: too many tiles");
  dim3 grid((unsigned)tiles);
  int rc = d->b_is_nk ? dispatch_epi<true>(p, epi, a_drop, grid, (hipStream_t)stream)
                      : dispatch_epi<false>(p, epi, a_drop, grid, (hipStream_t)stream);
  if (rc != B4R_OK) return rc;
  B4R_CHECK_LAUNCH("b4r_gemm_f32");
  return B4R_OK;
}

// C[M,N] = A.B with the K range split over `splits` workgroups per tile: partial products go to slabs in `scratch`
// (>= splits*M*N floats) and are summed in a fixed order.  Used where M*N is small and K is long (dT = dlogits.E, K = V).
bool b4r_gemm_rx_splitk_supported(const b4r_gemm_desc* d, int k_pad_ok);
int b4r_gemm_rx_splitk_launch(const b4r_gemm_desc* d, int splits, float* slabs, int* slabs_used, hipStream_t stream);

int b4r_gemm_f32_splitk(const b4r_gemm_desc* d, int splits, float* scratch, int k_pad_ok, hipStream_t stream) {
  B4R_CHECK_ARG(d && scratch && d->epilogue == B4R_EPI_NONE && !d->a_dropout, B4R_E_BADARG, "gemm_splitk: plain product only");
  if (splits > 1 && g_gemm_mode == B4R_GEMM_BF16X3 && b4r_gemm_rx_splitk_supported(d, k_pad_ok)) {
    int used = 0;
    int rc = b4r_gemm_rx_splitk_launch(d, splits, scratch, &used, stream);
    if (rc) return rc;
    return b4r_launch_slab_reduce(scratch, used, d->M, d->N, d->C, d->ldc, 0, stream);
  }
  if (splits <= 1) return b4r_gemm_f32(d, (b4r_stream_t)stream);
  GemmP p;
  p.A = d->A; p.B = d->B; p.C = scratch; p.bias = nullptr; p.C2 = nullptr; p.R = nullptr;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->N; p.ldc2 = 0; p.ldr = 0;
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.tiles_n = b4r_cdiv(d->N, BN);
  p.k_chunk = b4r_cdiv(b4r_cdiv(d->K, splits), BK) * BK;
  p.split_stride = (int64_t)d->M * d->N;
  p.a_vec = (b4r_aligned16(d->A) && (d->lda % 4 == 0)) ? 1 : 0;
  p.b_vec = (b4r_aligned16(d->B) && (d->ldb % 4 == 0)) ? 1 : 0;
  p.qscale = 1.f; p.qcols = 0;
  p.drop = b4r_make_drop(nullptr, 0, 0.f, 0);
  const int64_t tiles = (int64_t)b4r_cdiv(d->M, BM) * p.tiles_n;
  dim3 grid((unsigned)tiles, (unsigned)splits);
  if (d->b_is_nk)
    hipLaunchKernelGGL((gemm_kernel<true, B4R_EPI_NONE, false>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((gemm_kernel<false, B4R_EPI_NONE, false>), grid, dim3(256), 0, stream, p);
  B4R_CHECK_LAUNCH("gemm_splitk");
  return b4r_launch_slab_reduce(scratch, splits, d->M, d->N, d->C, d->ldc, 0, stream);
}

extern "C" int64_t b4r_gemm_tn_scratch_floats(int32_t R, int32_t Mo, int32_t No) {
  if (R <= 0 || Mo <= 0 || No <= 0) return 0;
  const int S = tn_split(R, Mo, No);
  const int64_t a = (int64_t)S * Mo * No + (int64_t)S * No + (int64_t)S * Mo;
  const int64_t b = b4r_gemm_rx_tn_scratch_floats(R, Mo, No);
  return a > b ? a : b;  // large enough for either arithmetic mode
}

extern "C" int b4r_gemm_tn_dgrad_supported(const b4r_gemm_tn_desc* d) {
  return (d != nullptr && g_gemm_mode == B4R_GEMM_BF16X3 && d->Mo >= 64 && d->Mo % 64 == 0 && d->No == 64 && d->R > 0 && d->A &&
          d->B && d->dgrad_w && d->dgrad_out && d->dgrad_ldw >= 64 && d->dgrad_ldw % 4 == 0 && b4r_aligned16(d->dgrad_w) &&
          d->dgrad_ldo >= d->Mo && (d->dgrad_gelu_pre == nullptr || d->dgrad_ldg >= d->Mo) && b4r_gemm_rx_tn_supported(d)) ? 1 : 0;
}

extern "C" int b4r_gemm_tn_f32(const b4r_gemm_tn_desc* d, float* scratch, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr && scratch != nullptr, B4R_E_BADARG, "b4r_gemm_tn_f32: null argument");
  B4R_CHECK_ARG(d->A && d->B && d->out, B4R_E_BADARG, "b4r_gemm_tn_f32: null operand");
  B4R_CHECK_ARG(d->R > 0 && d->Mo > 0 && d->No > 0, B4R_E_SHAPE, "b4r_gemm_tn_f32: bad shape");
  B4R_CHECK_ARG(d->lda >= d->Mo && d->ldb >= d->No && d->ldo >= d->No, B4R_E_SHAPE, "b4r_gemm_tn_f32: bad leading dimension");
  b4r_timing_detail("R=%d Mo=%d No=%d%s", d->R, d->Mo, d->No, d->dgrad_out ? " +dgrad" : "");
  if (d->dgrad_out != nullptr) {
    B4R_CHECK_ARG(d->dgrad_w != nullptr, B4R_E_BADARG, "b4r_gemm_tn_f32: dgrad_out needs dgrad_w");
    B4R_CHECK_ARG(b4r_gemm_tn_dgrad_supported(d), B4R_E_SHAPE,
                  "b4r_gemm_tn_f32: the fused input gradient needs No = 64, Mo %% 64 == 0, aligned operands and the bf16x3 mode "
                  "(b4r_gemm_tn_dgrad_supported)");
  }
  if (g_gemm_mode == B4R_GEMM_BF16X3 && b4r_gemm_rx_tn_supported(d)) return b4r_gemm_rx_tn_launch(d, scratch, (hipStream_t)stream);
  const int S = tn_split(d->R, d->Mo, d->No);
  TnP p;
  p.A = d->A; p.B = d->B; p.lda = d->lda; p.ldb = d->ldb;
  p.R = d->R; p.Mo = d->Mo; p.No = d->No;
  p.chunk = b4r_cdiv(b4r_cdiv(d->R, S), TK) * TK;
  p.slab = scratch;
  p.colsum_slab = d->colsum ? scratch + (int64_t)S * d->Mo * d->No : nullptr;
  p.colsum_a_slab = d->colsum_a ? scratch + (int64_t)S * d->Mo * d->No + (int64_t)S * d->No : nullptr;
  p.a_vec = (b4r_aligned16(d->A) && (d->lda % 4 == 0)) ? 1 : 0;
  p.b_vec = (b4r_aligned16(d->B) && (d->ldb % 4 == 0)) ? 1 : 0;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  const bool b_drop = d->b_dropout && p.drop.rng != nullptr;
  dim3 grid(b4r_cdiv(d->Mo, TB), b4r_cdiv(d->No, TB), S);
  if (b_drop)
    hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  B4R_CHECK_LAUNCH("b4r_gemm_tn_f32");
  return b4r_launch_slab_reduce_full(p.slab, S, d->Mo, d->No, d->out, d->ldo, d->accumulate, p.colsum_slab, d->colsum,
                                     p.colsum_a_slab, d->colsum_a, (hipStream_t)stream);
}

// two independent weight-gradient products, in one launch where the bf16x3 kernel takes both (d0 may drop its B operand, d1 not;
// no input-gradient tails), else one after the other: same results either way
int b4r_gemm_tn_pair(const b4r_gemm_tn_desc* d0, float* scratch0, const b4r_gemm_tn_desc* d1, float* scratch1, hipStream_t stream) {
  static const bool on = !(getenv("B4R_TN_PAIR") && atoi(getenv("B4R_TN_PAIR")) == 0);
  const bool plain = d0 && d1 && scratch0 && scratch1 && d0->A && d0->B && d0->out && d1->A && d1->B && d1->out && d0->R > 0 &&
                     d1->R > 0 && d0->Mo > 0 && d0->No > 0 && d1->Mo > 0 && d1->No > 0 && d0->lda >= d0->Mo && d0->ldb >= d0->No &&
                     d0->ldo >= d0->No && d1->lda >= d1->Mo && d1->ldb >= d1->No && d1->ldo >= d1->No;
  if (on && plain && g_gemm_mode == B4R_GEMM_BF16X3 && b4r_gemm_rx_tn_pair_supported(d0, d1))
    return b4r_gemm_rx_tn_pair_launch(d0, scratch0, d1, scratch1, stream);
  int rc = b4r_gemm_tn_f32(d0, scratch0, (b4r_stream_t)stream);
  if (rc) return rc;
  return b4r_gemm_tn_f32(d1, scratch1, (b4r_stream_t)stream);
}

