// Shared device helpers for the gfx950 (MI355X / CDNA4) BERT4Rec kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/b4r.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define B4R_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------------------------
void b4r_set_error(const char* fmt, ...);
#define B4R_CHECK_ARG(cond, code, ...)            \
  do {                                            \
    if (!(cond)) {                                \
      b4r_set_error(__VA_ARGS__);                 \
      return (code);                              \
    }                                             \
  } while (0)
// launch timing (b4r_timing_begin / b4r_timing_end, include/b4r.h): while a recording is active on the calling thread every
// checked launch is followed by a hipEvent on the recording's stream; both calls are no-ops otherwise
void b4r_timing_mark(const char* what);
void b4r_timing_detail(const char* fmt, ...);
#define B4R_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      b4r_set_error("%s: launch failed: %s", (what), hipGetErrorString(e__));    \
      return B4R_E_HIP;                                                          \
    }                                                                            \
    b4r_timing_mark(what);                                                       \
  } while (0)

static inline int b4r_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool b4r_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// deferred ordered reductions.  Every two-stage reduction of the backward pass (weight-gradient slabs, bias column
// sums, LayerNorm gamma/beta partials) can be queued instead of launched: b4r_backward then sums ALL of them with one
// kernel instead of ~18 tiny launches (each boundary costs ~1.5-2 us on this GPU whatever the kernel does).
// ---------------------------------------------------------------------------------------------
struct B4rReduceJob {
  const float* slab; const float* cslab; const float* caslab;
  float* out; float* colsum; float* colsum_a;
  int S, Mo, No, ldo, accumulate;
  // optional 64-bit fixed-point addend of the [Mo, No] matrix (units of 2^-36: the item-table scatter of b4r_embed_grads):
  // element e receives fix[e] + sum over fix_slots of fix_hot[slot][e] (e < fix_hot_elems) on top of the slab sum
  const long long* fix = nullptr; const long long* fix_hot = nullptr;
  int fix_hot_elems = 0, fix_slots = 0;
  const int* fix_poison = nullptr;   // != 0: a contribution was not finite or out of the fixed-point range -> every element becomes NaN
};
constexpr int B4R_MAX_REDUCE_JOBS = 40;
struct B4rReduceQueue {
  B4rReduceJob jobs[B4R_MAX_REDUCE_JOBS];
  int n;
};
void b4r_reduce_queue_begin(B4rReduceQueue* q);                 // queue reductions issued by this thread from now on
// launch them all (one kernel) and stop queueing.  sq_partial (optional, sq_cap floats): one sum of squares of the stored values per
// workgroup, *sq_np of them (0 when the launch has more workgroups than sq_cap); *covered: the number of elements the jobs write
int b4r_reduce_queue_flush(hipStream_t stream, float* sq_partial = nullptr, int sq_cap = 0, int* sq_np = nullptr,
                           int64_t* covered = nullptr);
bool b4r_reduce_queue_push(const B4rReduceJob& job);             // false: no queue active (caller reduces immediately)
// give the queued job that writes `out` a fixed-point addend; false: no queue, or no queued job writes `out`
bool b4r_reduce_queue_attach_fixed(const float* out, const long long* fix, const long long* fix_hot, int fix_hot_elems, int fix_slots,
                                   const int* fix_poison = nullptr);
int b4r_launch_reduce_job(const B4rReduceJob& job, hipStream_t stream);   // queued when a queue is active, else launched

// the deferred merge of the logits-free masked-LM head's forward (b4r_head_merge.h), as the hosts pass it around
struct B4rHeadMerge {
  const float* part; int slices, M, V;
  const float* T; const float* E; const float* bias; const int64_t* y;
  float* row_out; float* lse_out; int32_t* ylab;
};

// dropout sites (stream ids of the counter-hash RNG); restated in oracle/bert4rec_oracle.py
#define B4R_STREAM_EMB 0u
#define B4R_STREAM_ATTN_PROBS(layer) (1u + 4u * (uint32_t)(layer))
#define B4R_STREAM_ATTN_OUT(layer) (2u + 4u * (uint32_t)(layer))
#define B4R_STREAM_FFN_OUT(layer) (3u + 4u * (uint32_t)(layer))

// ---------------------------------------------------------------------------------------------
// counter-hash dropout.  keep(idx) is a pure function of (seed, step, stream, idx) so forward and backward
// regenerate the same mask without storing it.  Restated in oracle/bert4rec_oracle.py::dropout_keep_mask.
// One 2-round integer hash serves the GROUP of 4 consecutive elements idx>>2: the hash and one xorshift32 step of it give
// four 16-bit uniforms (v_mul_lo_u32 is a quarter-rate instruction and the hash was the dominant VALU cost of the
// attention kernels).  The drop probability is floor(rate * 65536) / 65536.  Element indices are laid out so that the 4
// elements a lane holds are one group: row-major [rows, N] tensors with N % 4 == 0, and attention probabilities indexed
// ((b*heads + h) * L + query) * 256 + key (B4R_ATTN_PITCH).
// ---------------------------------------------------------------------------------------------
constexpr int B4R_ATTN_PITCH = 256;

struct DropArgs {
  const uint32_t* rng;  // device: rng[0] = seed, rng[1] = step (low 32 bits); nullptr => dropout disabled
  uint32_t stream;      // dropout site id (B4R_STREAM_*)
  uint32_t thr;         // drop if u16 < thr;  thr = (uint32)(rate * 2^16)
  float scale;          // 1 / (1 - rate)
};

static inline DropArgs b4r_make_drop(const uint32_t* rng, uint32_t stream, float rate, int training) {
  DropArgs d;
  d.rng = nullptr; d.stream = stream; d.thr = 0; d.scale = 1.0f;
  if (training && rate > 0.0f && rng != nullptr) {
    d.rng = rng;
    d.thr = (uint32_t)((double)rate * 65536.0);
    d.scale = 1.0f / (1.0f - rate);
  }
  return d;
}

__device__ __forceinline__ uint32_t b4r_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// uniform in the OPEN interval (0, 1) from the top 23 bits of a hash word: (k + 0.5) / 2^23 is exact in fp32 for every k < 2^23, so
// the largest value is 1 - 2^-24 < 1 (24 bits would round (2^24 - 0.5) / 2^24 up to 1.0f: `u < rate` would then fail at rate = 1).
// Restates python's random.random() in [0, 1) (dataloader_utils.py:245-253) and numpy's uniform for the Gumbel keys.
__host__ __device__ __forceinline__ float b4r_uniform23(uint32_t h) {
  return ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

struct DropCtx {
  uint32_t seed, key, thr;
  float scale;
  bool on;
};

__device__ __forceinline__ DropCtx b4r_drop_ctx(const DropArgs& a) {
  DropCtx c;
  c.on = (a.rng != nullptr) && (a.thr != 0);
  c.thr = a.thr; c.scale = a.scale; c.seed = 0; c.key = 0;
  if (c.on) {
    c.seed = a.rng[0];
    c.key = b4r_hash32(a.stream * 0x9E3779B9u + a.rng[1]);
  }
  return c;
}

// the two 32-bit words of group g: uniforms of elements 4g+0 / 4g+1 are the low / high half of h1, 4g+2 / 4g+3 of h2
__device__ __forceinline__ void b4r_group_hash(const DropCtx& c, uint64_t g, uint32_t& h1, uint32_t& h2) {
  const uint32_t lo = (uint32_t)g, hi = (uint32_t)(g >> 32);
  uint32_t h = b4r_hash32(lo ^ c.seed);
  h = b4r_hash32((h ^ hi) + c.key);
  h1 = h;
  h ^= h << 13; h ^= h >> 17; h ^= h << 5;
  h2 = h;
}

// the four keep decisions of a group as predicates (lane masks): selecting on them directly costs one v_cndmask per element,
// testing the packed bits again costs an and + compare more
struct B4rKeep4 {
  bool k[4];
  __device__ __forceinline__ uint32_t bits() const { return (k[0] ? 1u : 0u) | (k[1] ? 2u : 0u) | (k[2] ? 4u : 0u) | (k[3] ? 8u : 0u); }
};
__device__ __forceinline__ B4rKeep4 b4r_keep4p(const DropCtx& c, uint64_t idx4) {
  uint32_t h1, h2;
  b4r_group_hash(c, idx4 >> 2, h1, h2);
  B4rKeep4 r;
  r.k[0] = (h1 & 0xFFFFu) >= c.thr; r.k[1] = (h1 >> 16) >= c.thr;
  r.k[2] = (h2 & 0xFFFFu) >= c.thr; r.k[3] = (h2 >> 16) >= c.thr;
  return r;
}
// bit e of the result = keep(idx4 + e); idx4 must be a multiple of 4
__device__ __forceinline__ uint32_t b4r_keep4(const DropCtx& c, uint64_t idx4) { return b4r_keep4p(c, idx4).bits(); }

__device__ __forceinline__ bool b4r_keep(const DropCtx& c, uint64_t idx) {
  uint32_t h1, h2;
  b4r_group_hash(c, idx >> 2, h1, h2);
  const uint32_t w = (idx & 2) ? h2 : h1;
  const uint32_t u = (idx & 1) ? (w >> 16) : (w & 0xFFFFu);
  return u >= c.thr;
}

// x -> dropout(x)
__device__ __forceinline__ float b4r_drop(const DropCtx& c, float x, uint64_t idx) {
  if (!c.on) return x;
  return b4r_keep(c, idx) ? x * c.scale : 0.0f;
}

// 4 consecutive elements idx .. idx+3 of one row.  One hash when they are one group (idx % 4 == 0, which holds at every
// call site whenever the row length is a multiple of 4), the per-element path otherwise; same decisions either way.
__device__ __forceinline__ f32x4 b4r_drop4(const DropCtx& c, f32x4 x, uint64_t idx) {
  if (!c.on) return x;
  if ((idx & 3) == 0) {
    const B4rKeep4 k = b4r_keep4p(c, idx);
    const f32x4 xs = x * c.scale;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = k.k[e] ? xs[e] : 0.0f;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = b4r_keep(c, idx + e) ? x[e] * c.scale : 0.0f;
  }
  return x;
}

// raise a kernel's dynamic-LDS limit once per (kernel, size): hipFuncSetAttribute costs ~10 us of host time per call,
// and the attention kernels (58 KB) would pay it at every launch
#include <mutex>
#include <unordered_map>
static inline int b4r_raise_lds(const void* kernel, size_t bytes, const char* who) {
  if (bytes <= 48 * 1024) return B4R_OK;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> raised;
  std::lock_guard<std::mutex> lock(mu);
  auto it = raised.find(kernel);
  if (it != raised.end() && it->second >= bytes) return B4R_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) { b4r_set_error("%s: cannot raise the LDS limit to %zu: %s", who, bytes, hipGetErrorString(e)); return B4R_E_HIP; }
  raised[kernel] = bytes;
  return B4R_OK;
}

// ---------------------------------------------------------------------------------------------
// split precision: x = hi + lo with hi = bf16(x), lo = bf16(x - hi).  Written on packed pairs: hipcc's own lowering of
// convert(x - convert(hi)) converts every element a second time on its own (32 VALU instructions per 8 elements, 20 here)
// ---------------------------------------------------------------------------------------------
typedef __bf16 b4r_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4r_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b4r_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b4r_f32x8 __attribute__((ext_vector_type(8)));
typedef float b4r_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int b4r_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int b4r_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void b4r_split_pair(float a, float b, uint32_t& hw, uint32_t& lw) {
  const b4r_bf16x2 h = __builtin_convertvector((b4r_f32x2){a, b}, b4r_bf16x2);
  hw = __builtin_bit_cast(uint32_t, h);
  const float h0 = __builtin_bit_cast(float, hw << 16), h1 = __builtin_bit_cast(float, hw & 0xFFFF0000u);
  lw = __builtin_bit_cast(uint32_t, __builtin_convertvector((b4r_f32x2){a - h0, b - h1}, b4r_bf16x2));
}
__device__ __forceinline__ void b4r_split8(const b4r_f32x8 x, b4r_bf16x8& hi, b4r_bf16x8& lo) {
  b4r_u32x4 hw, lw;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t h, l;
    b4r_split_pair(x[2 * j], x[2 * j + 1], h, l);
    hw[j] = h; lw[j] = l;
  }
  hi = __builtin_bit_cast(b4r_bf16x8, hw);
  lo = __builtin_bit_cast(b4r_bf16x8, lw);
}
__device__ __forceinline__ void b4r_split4(const f32x4 x, b4r_bf16x4& hi, b4r_bf16x4& lo) {
  b4r_u32x2 hw, lw;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    uint32_t h, l;
    b4r_split_pair(x[2 * j], x[2 * j + 1], h, l);
    hw[j] = h; lw[j] = l;
  }
  hi = __builtin_bit_cast(b4r_bf16x4, hw);
  lo = __builtin_bit_cast(b4r_bf16x4, lw);
}

// ---------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float b4r_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx [0.5 x (1+erf(x/sqrt2))] = 0.5 (1+erf(x/sqrt2)) + x * exp(-x^2/2) / sqrt(2 pi)
__device__ __forceinline__ float b4r_gelu_grad(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Cheaper variants for the split-precision kernels, whose epilogues were VALU-bound on erff (ocml: ~40 instructions, two
// code paths): Abramowitz & Stegun 7.1.26, erf(u) = 1 - (a1 t + .. + a5 t^5) exp(-u^2), t = 1/(1 + p|u|).  In fp32
// arithmetic |error| <= 6e-7 on erf, 5e-7 on gelu and 3.2e-7 on its derivative (checked on 2M points of [-8, 8]); the
// exponential is shared with the Gaussian term of the derivative.
__device__ __forceinline__ float b4r_erf_as(float u, float& e) {
  const float a = fabsf(u);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
  e = __expf(-a * a);
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  return copysignf(fmaf(-q * t, e, 1.0f), u);
}
__device__ __forceinline__ float b4r_gelu_fast(float x) {
  float e;
  return 0.5f * x * (1.0f + b4r_erf_as(x * 0.70710678118654752440f, e));
}
__device__ __forceinline__ float b4r_gelu_grad_fast(float x) {
  float e;
  const float er = b4r_erf_as(x * 0.70710678118654752440f, e);
  return fmaf(x * 0.39894228040143267794f, e, 0.5f * (1.0f + er));
}

// The largest additive key mask of one sequence: 0 when at least one key is valid, -1e9 when the whole row of input_mask is zero.
// Keras adds (1 - mask) * -1e9 in fp32; on a fully masked sequence every score rounds to -1e9 (ulp 64) and the softmax is exactly
// uniform, but log-sum-exp = -1e9 + log(L) cannot hold its log(L) in fp32 -- a backward that recomputes exp(score - lse) would see
// probability 1 for every key.  All attention kernels therefore store lse relative to this shift,
//     lse = (max - amax) + log(sum) ,   p = exp(((score + mask) - amax) - lse) ,
// which changes nothing when a key is valid (amax = 0) and keeps the rounding of (score + mask) that makes the row uniform.
// Contains a barrier: every thread of the workgroup calls it, with block-uniform arguments.
__device__ __forceinline__ float b4r_seq_amax(const int64_t* mask_row, int L) {
  int any = 0;
  for (int k = threadIdx.x; k < L; k += blockDim.x) any |= (mask_row[k] != 0) ? 1 : 0;
  return __syncthreads_or(any) ? 0.0f : -1e9f;
}

__device__ __forceinline__ float b4r_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float b4r_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
