// Dense layers for K <= 64 (every projection whose reduction dimension is the hidden size of the 64-wide configurations:
// QKV, attention output, FFN-in, the MLM transform and the tied vocabulary projection, and the input gradients that
// reduce over H) on the bf16 matrix cores with a 3-term split ("bf16x3"):
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
  float qscale; int qcols;
  DropArgs drop;
};

__device__ __forceinline__ void split8(const f32x8 x, bf16x8& hi, bf16x8& lo) {
  hi = __builtin_convertvector(x, bf16x8);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x8), bf16x8);
}

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

constexpr bool epi_has_bias(int e) {
  return e == B4R_EPI_BIAS || e == B4R_EPI_BIAS_QSCALE || e == B4R_EPI_BIAS_GELU || e == B4R_EPI_BIAS_DROP_RES ||
         e == B4R_EPI_BIAS_TANH;
}
constexpr bool epi_has_r(int e) { return e == B4R_EPI_BIAS_DROP_RES || e == B4R_EPI_GELU_BWD || e == B4R_EPI_ADD_RES; }

// the wave's 32 x K strip of A, split into hi/lo fragments (row must be valid: M % 32 == 0 and the wave is live)
template <bool A_DROP, int NKB>
__device__ __forceinline__ void load_a_strip(const RxP& p, const DropCtx& dctx, int row, int h, bf16x8 (&ah)[NKB], bf16x8 (&al)[NKB]) {
  const float* arow = p.A + (int64_t)row * p.lda + 8 * h;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    f32x8 x = load8_contig(arow + 16 * kb);
    if (A_DROP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = b4r_drop(dctx, x[j], (uint64_t)row * (uint64_t)p.K + (uint64_t)(16 * kb + 8 * h + j));
    }
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

template <int EPI>
__device__ __forceinline__ void epilogue_tile(const RxP& p, const DropCtx& dctx, const f32x16& acc, const f32x4 bv,
                                              float* stage, int m0, int n0, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) stage[((reg & 3) + 8 * (reg >> 2) + 4 * h) * ST_LD + r] = acc[reg];
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes have landed; no other wave touches the tile
  const int c4 = (lane & 7) * 4, rsub = lane >> 3;
  const int col = min(n0 + c4, p.n_store - 4);
  f32x4 vin[4], rr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lrow = rsub + 8 * i;
    vin[i] = *reinterpret_cast<const f32x4*>(&stage[lrow * ST_LD + (col - n0)]);
    if (epi_has_r(EPI)) rr[i] = *reinterpret_cast<const f32x4*>(p.R + (int64_t)(m0 + lrow) * p.ldr + col);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + rsub + 8 * i;
    f32x4 o, o2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = vin[i][e];
      float y;
      if (EPI == B4R_EPI_NONE) y = a;
      else if (EPI == B4R_EPI_BIAS) y = a + bv[e];
      else if (EPI == B4R_EPI_BIAS_QSCALE) y = (a + bv[e]) * ((col + e < p.qcols) ? p.qscale : 1.0f);
      else if (EPI == B4R_EPI_BIAS_GELU) { o2[e] = a + bv[e]; y = b4r_gelu(o2[e]); }
      else if (EPI == B4R_EPI_BIAS_DROP_RES)
        y = rr[i][e] + b4r_drop(dctx, a + bv[e], (uint64_t)row * (uint64_t)p.N + (uint64_t)(col + e));
      else if (EPI == B4R_EPI_GELU_BWD) y = a * b4r_gelu_grad(rr[i][e]);
      else if (EPI == B4R_EPI_ADD_RES) y = a + rr[i][e];
      else y = tanhf(a + bv[e]);
      o[e] = y;
    }
    *reinterpret_cast<f32x4*>(p.C + (int64_t)row * p.ldc + col) = o;
    if (EPI == B4R_EPI_BIAS_GELU) *reinterpret_cast<f32x4*>(p.C2 + (int64_t)row * p.ldc2 + col) = o2;
  }
}

// contract (b4r_gemm_rx_supported): K = 16*NKB; M % 32 == 0; all operands 16-byte aligned with ld % 4 == 0;
// columns [N, n_store) of C (and C2) may be written, of R may be read

// ---- B as [K,N]: register operands, no barriers ----------------------------------------------------------------------
template <int EPI, bool A_DROP, int NKB>
__global__ __launch_bounds__(256) void rx_gemm_kn_kernel(RxP p) {
  extern __shared__ __attribute__((aligned(16))) float s_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int mblock = blockIdx.x / p.n_splits, split = blockIdx.x % p.n_splits;
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

  for (int s = s_begin; s < s_end; ++s) {
    const int n0 = s * 32;
    bf16x8 bh[NKB], bl[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) split8(braw[kb], bh[kb], bl[kb]);
    const f32x4 bv = bias_next;
    // unconditional look-ahead (the last step re-requests its own tile: cheaper than a branch, see header)
    const int n_next = min(n0 + 32, (s_end - 1) * 32);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) braw[kb] = load_b_raw(n_next, kb);
    bias_next = load_bias4<EPI>(p, n_next, c4);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc = mfma3(ah[kb], al[kb], bh[kb], bl[kb], acc);
    epilogue_tile<EPI>(p, dctx, acc, bv, stage, m0, n0, lane);
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
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int mblock = blockIdx.x / p.n_splits, split = blockIdx.x % p.n_splits;
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
      const bf16x4 hi = __builtin_convertvector(raw[i], bf16x4);
      const bf16x4 lo = __builtin_convertvector(raw[i] - __builtin_convertvector(hi, f32x4), bf16x4);
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
  __syncthreads();

  for (int s = s_begin; s < s_end; ++s) {
    const int n0 = s * 32, cur = (s - s_begin) & 1;
    const int n_next = min(n0 + 32, (s_end - 1) * 32);
    fetch_b(n_next);                                   // in flight under the MFMAs and older than this step's stores
    const f32x4 bias_next = load_bias4<EPI>(p, n_next, c4);
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
    if (live) epilogue_tile<EPI>(p, dctx, acc, bv, stage, m0, n0, lane);
    stash_b(cur ^ 1);                                  // the other buffer: nobody reads it during this step
    bv = bias_next;
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
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128 + wave * 32;
  const int nb = blockIdx.y * (32 * NT);            // first column of this workgroup's pass over N
  const bool live = m0 < p.M;
  const int nchunks = p.K / 64;
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
        const bf16x4 hi = __builtin_convertvector(braw[i], bf16x4);
        const bf16x4 lo = __builtin_convertvector(braw[i] - __builtin_convertvector(hi, f32x4), bf16x4);
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

  fetch_a(0);
  fetch_b(0);
  stash_b(0);
  if (B_NK) __syncthreads();

  for (int c = 0; c < nchunks; ++c) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      f32x8 x = araw[kb];
      if (A_DROP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = b4r_drop(dctx, x[j], (uint64_t)arow * (uint64_t)p.K + (uint64_t)(64 * c + 16 * kb + 8 * h + j));
      }
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
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (nb + 32 * j < p.N) epilogue_tile<EPI>(p, dctx, acc[j], load_bias4<EPI>(p, nb + 32 * j, c4), stage, m0, nb + 32 * j, lane);
    }
  }
}

template <bool B_NK, int EPI, bool A_DROP>
void launch_kloop(const RxP& p, hipStream_t s) {
  const int nt = b4r_cdiv(p.N, 32);
  dim3 grid((unsigned)b4r_cdiv(p.M, 128), (unsigned)(nt <= 2 ? 1 : b4r_cdiv(nt, 4)));
  if (nt <= 2) {
    const size_t lds = STAGE_FLOATS * sizeof(float) + (B_NK ? 2 * 2 * 2 * 32 * 36 * 4 : 0);
    hipLaunchKernelGGL((rx_gemm_kloop_kernel<B_NK, EPI, A_DROP, 2>), grid, dim3(256), lds, s, p);
  } else {
    const size_t lds = STAGE_FLOATS * sizeof(float) + (B_NK ? 2 * 2 * 4 * 32 * 36 * 4 : 0);
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)rx_gemm_kloop_kernel<B_NK, EPI, A_DROP, 4>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((rx_gemm_kloop_kernel<B_NK, EPI, A_DROP, 4>), grid, dim3(256), lds, s, p);
  }
}

template <bool B_NK, int EPI, bool A_DROP>
void launch_rx2(const RxP& p, dim3 grid, hipStream_t s) {
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
// k is the ROW index of both operands, so every fragment is 8 coalesced dword loads (32 lanes = one 128-byte line).  One
// wave = one (64 x 64 output tile, slice of R) work item accumulating in registers; the next 16 rows are requested before
// the MFMAs of the current 16; partial tiles go to slabs and are summed in a fixed order (bitwise reproducible).  Bias
// gradients (column sums of B, or of A for the tied-table / output-bias pair) ride along on the loaded values.
struct RxTnP {
  const float* A; const float* B; float* slab; float* colsum_slab; float* colsum_a_slab;
  int lda, ldb;
  int R, Mo, No;
  int tiles_i, tiles_j, S, chunk;  // chunk: rows per slice, multiple of 16
  DropArgs drop;
};

template <bool B_DROP>
__global__ __launch_bounds__(256) void rx_gemm_tn_kernel(RxTnP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  __shared__ float s_red[3 * 64 * 64 + 3 * 256];   // partial tiles + column sums of waves 1..3
  const int64_t item = blockIdx.x;                  // one workgroup = one (tile, slice); its 4 waves split the slice
  const int z = (int)(item / ((int64_t)p.tiles_i * p.tiles_j));
  const int t = (int)(item % ((int64_t)p.tiles_i * p.tiles_j));
  const int ti = t / p.tiles_j, tj = t % p.tiles_j;
  const int i0 = ti * 64, j0 = tj * 64;
  const int sub = p.chunk / 4;                      // multiple of 16
  const int r_begin = min(p.R, z * p.chunk + wave * sub), r_end = min(p.R, r_begin + sub);   // multiples of 16
  const bool do_cs = p.colsum_slab != nullptr && ti == 0;
  const bool do_csa = p.colsum_a_slab != nullptr && tj == 0;
  DropCtx dctx = b4r_drop_ctx(p.drop);
  // out-of-range columns are clamped (valid memory, results never stored)
  const int ci0 = min(i0 + r, p.Mo - 1), ci1 = min(i0 + 32 + r, p.Mo - 1);
  const int cj0 = min(j0 + r, p.No - 1), cj1 = min(j0 + 32 + r, p.No - 1);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  float cs0 = 0.f, cs1 = 0.f, csa0 = 0.f, csa1 = 0.f;

  f32x8 xa0, xa1, xb0, xb1;
  auto fetch = [&](int k0) {
    const int kk = k0 + 8 * h;
    xa0 = load8_strided(p.A + (int64_t)kk * p.lda + ci0, p.lda);
    xa1 = load8_strided(p.A + (int64_t)kk * p.lda + ci1, p.lda);
    xb0 = load8_strided(p.B + (int64_t)kk * p.ldb + cj0, p.ldb);
    xb1 = load8_strided(p.B + (int64_t)kk * p.ldb + cj1, p.ldb);
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int k0 = r_begin; k0 < r_end; k0 += 16) {
    f32x8 ya0 = xa0, ya1 = xa1, yb0 = xb0, yb1 = xb1;
    fetch(min(k0 + 16, r_end - 16));   // unconditional look-ahead
    if (B_DROP) {
      const int kk = k0 + 8 * h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        yb0[j] = b4r_drop(dctx, yb0[j], (uint64_t)(kk + j) * (uint64_t)p.No + (uint64_t)cj0);
        yb1[j] = b4r_drop(dctx, yb1[j], (uint64_t)(kk + j) * (uint64_t)p.No + (uint64_t)cj1);
      }
    }
    if (do_cs) {
      cs0 += ((yb0[0] + yb0[1]) + (yb0[2] + yb0[3])) + ((yb0[4] + yb0[5]) + (yb0[6] + yb0[7]));
      cs1 += ((yb1[0] + yb1[1]) + (yb1[2] + yb1[3])) + ((yb1[4] + yb1[5]) + (yb1[6] + yb1[7]));
    }
    if (do_csa) {
      csa0 += ((ya0[0] + ya0[1]) + (ya0[2] + ya0[3])) + ((ya0[4] + ya0[5]) + (ya0[6] + ya0[7]));
      csa1 += ((ya1[0] + ya1[1]) + (ya1[2] + ya1[3])) + ((ya1[4] + ya1[5]) + (ya1[6] + ya1[7]));
    }
    bf16x8 ah0, al0, ah1, al1, bh0, bl0, bh1, bl1;
    split8(ya0, ah0, al0); split8(ya1, ah1, al1); split8(yb0, bh0, bl0); split8(yb1, bh1, bl1);
    acc[0][0] = mfma3(ah0, al0, bh0, bl0, acc[0][0]);
    acc[0][1] = mfma3(ah0, al0, bh1, bl1, acc[0][1]);
    acc[1][0] = mfma3(ah1, al1, bh0, bl0, acc[1][0]);
    acc[1][1] = mfma3(ah1, al1, bh1, bl1, acc[1][1]);
  }

  // combine the 4 waves in a fixed order (wave 0 + 1 + 2 + 3), then one slab per workgroup
  if (wave > 0) {
    float* dst = s_red + (wave - 1) * 4096;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) dst[((a * 2 + b) * 16 + reg) * 64 + lane] = acc[a][b][reg];
    float* dc = s_red + 3 * 4096 + (wave - 1) * 256;
    dc[lane] = cs0; dc[64 + lane] = cs1; dc[128 + lane] = csa0; dc[192 + lane] = csa1;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float* src = s_red + w * 4096;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) acc[a][b][reg] += src[((a * 2 + b) * 16 + reg) * 64 + lane];
    const float* sc = s_red + 3 * 4096 + w * 256;
    cs0 += sc[lane]; cs1 += sc[64 + lane]; csa0 += sc[128 + lane]; csa1 += sc[192 + lane];
  }
  float* slab = p.slab + (int64_t)z * p.Mo * p.No;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = j0 + 32 * b + r;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = i0 + 32 * a + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (row < p.Mo && col < p.No) slab[(int64_t)row * p.No + col] = acc[a][b][reg];
      }
    }
  if (do_cs) {
    const float s0 = cs0 + __shfl_xor(cs0, 32, 64), s1 = cs1 + __shfl_xor(cs1, 32, 64);
    if (h == 0 && j0 + r < p.No) p.colsum_slab[(int64_t)z * p.No + j0 + r] = s0;
    if (h == 0 && j0 + 32 + r < p.No) p.colsum_slab[(int64_t)z * p.No + j0 + 32 + r] = s1;
  }
  if (do_csa) {
    const float s0 = csa0 + __shfl_xor(csa0, 32, 64), s1 = csa1 + __shfl_xor(csa1, 32, 64);
    if (h == 0 && i0 + r < p.Mo) p.colsum_a_slab[(int64_t)z * p.Mo + i0 + r] = s0;
    if (h == 0 && i0 + 32 + r < p.Mo) p.colsum_a_slab[(int64_t)z * p.Mo + i0 + 32 + r] = s1;
  }
}

int rx_tn_split(int R, int Mo, int No) {
  const int tiles = b4r_cdiv(Mo, 64) * b4r_cdiv(No, 64);
  int S = b4r_cdiv(192, tiles);   // workgroups of 4 waves; measured optimum on ML-1M shapes (96/192/384/768 tried)
  const int max_s = b4r_cdiv(R, 256);  // at least 64 rows per wave
  if (S > max_s) S = max_s;
  if (S > 256) S = 256;
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
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
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
  dim3 grid((unsigned)(mblocks * p.n_splits));
  int rc = d->b_is_nk ? dispatch_rx<true>(p, d->epilogue, a_drop, grid, stream)
                      : dispatch_rx<false>(p, d->epilogue, a_drop, grid, stream);
  if (rc != B4R_OK) return rc;
  B4R_CHECK_LAUNCH("b4r_gemm_f32 (bf16x3)");
  return B4R_OK;
}

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);

bool b4r_gemm_rx_tn_supported(const b4r_gemm_tn_desc* d) { return d->R % 16 == 0 && d->R >= 64; }

int64_t b4r_gemm_rx_tn_scratch_floats(int R, int Mo, int No) {
  const int S = rx_tn_split(R, Mo, No);
  return (int64_t)S * Mo * No + (int64_t)S * No + (int64_t)S * Mo;
}

int b4r_gemm_rx_tn_launch(const b4r_gemm_tn_desc* d, float* scratch, hipStream_t stream) {
  const int S = rx_tn_split(d->R, d->Mo, d->No);
  RxTnP p;
  p.A = d->A; p.B = d->B; p.lda = d->lda; p.ldb = d->ldb;
  p.R = d->R; p.Mo = d->Mo; p.No = d->No;
  p.tiles_i = b4r_cdiv(d->Mo, 64); p.tiles_j = b4r_cdiv(d->No, 64); p.S = S;
  p.chunk = b4r_cdiv(b4r_cdiv(d->R, S), 64) * 64;   // each of the 4 waves takes a quarter (multiple of 16 rows)
  p.slab = scratch;
  p.colsum_slab = d->colsum ? scratch + (int64_t)S * d->Mo * d->No : nullptr;
  p.colsum_a_slab = d->colsum_a ? scratch + (int64_t)S * d->Mo * d->No + (int64_t)S * d->No : nullptr;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  const bool b_drop = d->b_dropout && p.drop.rng != nullptr;
  const int64_t items = (int64_t)p.tiles_i * p.tiles_j * S;
  dim3 grid((unsigned)items);
  if (b_drop) hipLaunchKernelGGL((rx_gemm_tn_kernel<true>), grid, dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((rx_gemm_tn_kernel<false>), grid, dim3(256), 0, stream, p);
  B4R_CHECK_LAUNCH("b4r_gemm_tn_f32 (bf16x3)");
  return b4r_launch_slab_reduce_full(p.slab, S, d->Mo, d->No, d->out, d->ldo, d->accumulate, p.colsum_slab, d->colsum,
                                     p.colsum_a_slab, d->colsum_a, stream);
}
