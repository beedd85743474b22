// HBM-bound row kernels: embedding gather + positional add + LayerNorm, LayerNorm fwd/bwd, row gather / scatter-add,
// softmax cross-entropy with fused argmax metrics, global-norm + AdamW over the flat parameter buffer.
//
// Row layout: a row of width H = 4*LPR*NV floats is owned by LPR lanes, each holding NV float4 (16 B per lane per
// access: coalesced 16-B vector loads, guide G13); a 64-lane wave therefore owns 64/LPR rows, and row statistics are
// xor-shuffle reductions inside the LPR-lane group.
#include "b4r_common.h"
#include "b4r_head_merge.h"
#include "b4r_head32_pack.h"

int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);
int b4r_launch_slab_reduce(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                           hipStream_t stream);

namespace {

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// -----------------------------------------------------------------------------------------------------------
// LayerNorm forward (optionally fed by the embedding gather)
//   Keras non-fused LayerNormalization: mean, biased var, inv = rsqrt(var+eps)*gamma, y = x*inv + (beta - mean*inv)
// -----------------------------------------------------------------------------------------------------------
struct LnFwdP {
  const float* z;            // [rows,H]   (EMBED: unused)
  const int64_t* ids;        // EMBED: [rows] token ids
  const float* table;        // EMBED: [V,H]
  const float* pos_table;    // EMBED: [Lmax,H]
  int L, V;
  const float* gamma; const float* beta;
  float* y; float* mean; float* rstd;
  int rows, H;
  float eps;
  DropArgs drop;
};

template <int LPR, int NV, bool EMBED>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdP p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane % LPR, slot = lane / LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + slot;
  if (row >= p.rows) return;  // whole LPR-group exits together; shuffles below stay inside the group
  DropCtx dctx = b4r_drop_ctx(p.drop);

  f32x4 x[NV];
  float s = 0.f;
  if (EMBED) {
    int64_t id = p.ids[row];
    if (id < 0 || id >= p.V) id = 0;  // out-of-range ids read the PAD row instead of faulting
    const int l = (int)(row % p.L);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * LPR + sub) * 4;
      const f32x4 e = *reinterpret_cast<const f32x4*>(p.table + id * p.H + c);
      const f32x4 q = *reinterpret_cast<const f32x4*>(p.pos_table + (int64_t)l * p.H + c);
      x[v] = e + q;
    }
  } else {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * LPR + sub) * 4;
      x[v] = *reinterpret_cast<const f32x4*>(p.z + row * p.H + c);
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
  const float mean = group_sum<LPR>(s) / (float)p.H;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = x[v][e] - mean; q += d * d; }
  }
  const float var = group_sum<LPR>(q) / (float)p.H;
  const float rstd = rsqrtf(var + p.eps);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + sub) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float inv = rstd * g[e];
      o[e] = x[v][e] * inv + (b[e] - mean * inv);
    }
    o = b4r_drop4(dctx, o, (uint64_t)row * (uint64_t)p.H + (uint64_t)c);
    *reinterpret_cast<f32x4*>(p.y + row * p.H + c) = o;
  }
  if (sub == 0) {
    if (p.mean) p.mean[row] = mean;
    if (p.rstd) p.rstd[row] = rstd;
  }
}

// -----------------------------------------------------------------------------------------------------------
// LayerNorm backward.  dz = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma; column partials of dgamma/dbeta.
// EMBED: z is recomputed as E[id]+P[pos] and dy first goes back through the embedding dropout.
// -----------------------------------------------------------------------------------------------------------
struct LnBwdP {
  const float* dy; const float* z; const float* mean; const float* rstd; const float* gamma;
  const int64_t* ids; const float* table; const float* pos_table; int L, V;
  float* dz; float* partial;  // partial[gridDim.x][2][H]
  int rows, H;
  DropArgs drop;
  const float* gelu_pre;      // optional [rows,H]: dz is further multiplied by gelu'(gelu_pre) (dense+GELU before the LN)
  HeadMergeP merge;           // MERGE: dy is not read -- it is the masked-LM head's dT, merged here from the forward's V slices
};

// MERGE (LPR = 16, NV = 1: hidden size 64, the thread layout of head_combine_kernel): the LayerNorm of the masked-LM transform in a
// train step; the lanes of a row also write the row's loss scalars, lse and label (b4r_head_merge.h)
template <int LPR, int NV, bool EMBED, bool MERGE = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdP p) {
  constexpr int RPW = 64 / LPR;
  extern __shared__ float sred[];  // [4*RPW][2*H]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane % LPR, slot = lane / LPR;
  DropCtx dctx = b4r_drop_ctx(p.drop);

  f32x4 g4[NV], dg[NV], db[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    g4[v] = *reinterpret_cast<const f32x4*>(p.gamma + (v * LPR + sub) * 4);
    dg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    db[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int64_t rows_per_block = 4 * RPW;
  for (int64_t base = (int64_t)blockIdx.x * rows_per_block; base < p.rows; base += (int64_t)gridDim.x * rows_per_block) {
    const int64_t row = base + wave * RPW + slot;
    const bool live = row < p.rows;
    const int64_t rr = live ? row : 0;
    const float mean = p.mean[rr], rstd = p.rstd[rr];
    f32x4 xh[NV], gg[NV];
    float s1 = 0.f, s2 = 0.f;
    int64_t id = 0; int l = 0;
    if (EMBED) {
      id = p.ids[rr];
      if (id < 0 || id >= p.V) id = 0;
      l = (int)(rr % p.L);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = (v * LPR + sub) * 4;
      f32x4 d;
      if (MERGE) d = head_merge_row<2>(p.merge, (int)rr, sub);   // dead lanes repeat row 0 (same values, same addresses)
      else d = *reinterpret_cast<const f32x4*>(p.dy + rr * p.H + c);
      f32x4 zz;
      if (EMBED) {
        zz = *reinterpret_cast<const f32x4*>(p.table + id * p.H + c) +
             *reinterpret_cast<const f32x4*>(p.pos_table + (int64_t)l * p.H + c);
      } else {
        zz = *reinterpret_cast<const f32x4*>(p.z + rr * p.H + c);
      }
      if (EMBED) d = b4r_drop4(dctx, d, (uint64_t)rr * (uint64_t)p.H + (uint64_t)c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float de = d[e];
        if (!live) de = 0.f;
        const float xhat = (zz[e] - mean) * rstd;
        const float g = de * g4[v][e];
        xh[v][e] = xhat; gg[v][e] = g;
        s1 += g; s2 += g * xhat;
        dg[v][e] += de * xhat;
        db[v][e] += de;
      }
    }
    const float c1 = group_sum<LPR>(s1) / (float)p.H;
    const float c2 = group_sum<LPR>(s2) / (float)p.H;
    if (live) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = (v * LPR + sub) * 4;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (gg[v][e] - c1 - xh[v][e] * c2);
        if (p.gelu_pre) {
          const f32x4 pre = *reinterpret_cast<const f32x4*>(p.gelu_pre + row * p.H + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= b4r_gelu_grad(pre[e]);
        }
        *reinterpret_cast<f32x4*>(p.dz + row * p.H + c) = o;
      }
    }
  }
  // block-level column reduction of dgamma / dbeta in a fixed order
  const int srow = wave * RPW + slot;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + sub) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sred[srow * 2 * p.H + c + e] = dg[v][e];
      sred[srow * 2 * p.H + p.H + c + e] = db[v][e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * p.H; c += 256) {
    float s = 0.f;
    for (int r = 0; r < 4 * RPW; ++r) s += sred[r * 2 * p.H + c];
    p.partial[(int64_t)blockIdx.x * 2 * p.H + c] = s;
  }
}

int ln_bwd_grid(int rows, int H) {
  int lpr = H / 4; if (lpr > 64) lpr = 64;
  const int rpb = 4 * (64 / lpr);
  int g = b4r_cdiv(rows, rpb);
  if (g > 1024) g = 1024;   // 4 workgroups per CU: enough bytes in flight for an HBM-bound pass
  if (g < 1) g = 1;
  return g;
}

template <bool EMBED>
int launch_ln_fwd(const LnFwdP& p, hipStream_t s) {
  const int H = p.H;
#define LN_FWD_CASE(LPR_, NV_)                                                                              \
  {                                                                                                         \
    const int rpb = 4 * (64 / LPR_);                                                                        \
    hipLaunchKernelGGL((ln_fwd_kernel<LPR_, NV_, EMBED>), dim3(b4r_cdiv(p.rows, rpb)), dim3(256), 0, s, p); \
  }
  switch (H) {
    case 32: LN_FWD_CASE(8, 1) break;
    case 64: LN_FWD_CASE(16, 1) break;
    case 128: LN_FWD_CASE(32, 1) break;
    case 256: LN_FWD_CASE(64, 1) break;
    case 512: LN_FWD_CASE(64, 2) break;
    case 1024: LN_FWD_CASE(64, 4) break;
    default: b4r_set_error("layer norm: hidden size %d not supported (32,64,128,256,512,1024)", H); return B4R_E_SHAPE;
  }
#undef LN_FWD_CASE
  return B4R_OK;
}

template <bool EMBED>
int launch_ln_bwd(const LnBwdP& p, int grid, hipStream_t s) {
  const int H = p.H;
  if (p.merge.part != nullptr) {
    if (EMBED || H != 64) { b4r_set_error("layer norm backward: the head merge needs hidden size 64"); return B4R_E_SHAPE; }
    hipLaunchKernelGGL((ln_bwd_kernel<16, 1, false, true>), dim3(grid), dim3(256), (size_t)4 * 4 * 2 * H * sizeof(float), s, p);
    return B4R_OK;
  }
#define LN_BWD_CASE(LPR_, NV_)                                                                           \
  {                                                                                                      \
    const size_t sh = (size_t)4 * (64 / LPR_) * 2 * H * sizeof(float);                                   \
    hipLaunchKernelGGL((ln_bwd_kernel<LPR_, NV_, EMBED>), dim3(grid), dim3(256), sh, s, p);              \
  }
  switch (H) {
    case 32: LN_BWD_CASE(8, 1) break;
    case 64: LN_BWD_CASE(16, 1) break;
    case 128: LN_BWD_CASE(32, 1) break;
    case 256: LN_BWD_CASE(64, 1) break;
    case 512: LN_BWD_CASE(64, 2) break;
    case 1024: LN_BWD_CASE(64, 4) break;
    default: b4r_set_error("layer norm: hidden size %d not supported", H); return B4R_E_SHAPE;
  }
#undef LN_BWD_CASE
  return B4R_OK;
}

// -----------------------------------------------------------------------------------------------------------
// gather / scatter-add of rows, and the batch column-sum that yields the position-table gradient
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* src, int src_ld, const int64_t* idx,
                                                          int64_t idx_add_per, int per, int n, int H, float* dst) {
  const int h4 = H / 4;
  const int64_t total = (int64_t)n * h4;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int i = (int)(t / h4), c = (int)(t % h4) * 4;
    int64_t pos = idx[i];
    if (idx_add_per > 0) pos = pos < 0 ? 0 : (pos >= idx_add_per ? idx_add_per - 1 : pos);  // clamp into the group
    const int64_t r = pos + (int64_t)(i / per) * idx_add_per;
    *reinterpret_cast<f32x4*>(dst + (int64_t)i * H + c) = *reinterpret_cast<const f32x4*>(src + r * src_ld + c);
  }
}

constexpr int HOT_SLOTS = 64;
// one lane per float: a wave-instruction adds 64 consecutive dwords (256 contiguous bytes of one or a few rows), the
// shape at which global float atomics run at full rate (MI355X_MICROARCH.md 'Global float atomics')
__device__ __forceinline__ void scatter_add_rows_body(const float* src, const int64_t* idx, int64_t idx_add_per,
                                                      int per, int n, int H, float* dst, int dst_ld,
                                                      const int64_t* skip_if_zero, int64_t dst_rows,
                                                      int hot_rows, float* hot_slab, const int block, const int nblocks) {
  // hot_rows > 0: destination rows [0, hot_rows) (the special tokens PAD/MASK/UNK of the item table: [MASK] alone is
  // ~20 % of all tokens) are first summed in LDS and leave the workgroup once, instead of thousands of global atomics
  // queueing on the same 256 bytes.  Even one flush per workgroup is 1024 same-address atomics per float, which the L2
  // serialises (that was most of this kernel's 39 us): with hot_slab the per-workgroup sums go to HOT_SLOTS slabs
  // [HOT_SLOTS][hot_rows*H] (workgroup % HOT_SLOTS: 16 atomics per address instead of 1024) that a slab reduction adds
  // to dst afterwards.
  extern __shared__ float s_hot[];
  for (int k = threadIdx.x; k < hot_rows * H; k += 256) s_hot[k] = 0.f;
  if (hot_rows > 0) __syncthreads();
  const int64_t total = (int64_t)n * H;
  for (int64_t t = (int64_t)block * 256 + threadIdx.x; t < total; t += (int64_t)nblocks * 256) {
    const int i = (int)(t / H), c = (int)(t % H);
    if (skip_if_zero != nullptr && skip_if_zero[i] == 0) continue;
    int64_t pos = idx[i];
    if (idx_add_per > 0) pos = pos < 0 ? 0 : (pos >= idx_add_per ? idx_add_per - 1 : pos);
    const int64_t r = pos + (int64_t)(i / per) * idx_add_per;
    if (r < 0 || r >= dst_rows) continue;
    const float v = src[(int64_t)i * H + c];
    if (r < hot_rows) atomicAdd(&s_hot[(int)r * H + c], v);
    else atomicAdd(dst + r * dst_ld + c, v);
  }
  if (hot_rows > 0) {
    __syncthreads();
    for (int k = threadIdx.x; k < hot_rows * H; k += 256) {
      const float v = s_hot[k];
      if (v == 0.f) continue;
      if (hot_slab) atomicAdd(hot_slab + (int64_t)(block % HOT_SLOTS) * hot_rows * H + k, v);
      else atomicAdd(dst + (int64_t)(k / H) * dst_ld + (k % H), v);
    }
  }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* src, const int64_t* idx, int64_t idx_add_per,
                                                               int per, int n, int H, float* dst, int dst_ld,
                                                               const int64_t* skip_if_zero, int64_t dst_rows,
                                                               int hot_rows, float* hot_slab) {
  scatter_add_rows_body(src, idx, idx_add_per, per, n, H, dst, dst_ld, skip_if_zero, dst_rows, hot_rows, hot_slab, (int)blockIdx.x,
                        (int)gridDim.x);
}

// partial[s][l][c] = sum over b in slice s of x[(b*L + l)*H + c]
__device__ __forceinline__ void batch_colsum_body(const float* x, int B, int L, int H, int bchunk, float* partial, const int bx, const int by) {
  const int h4 = H / 4;
  const int64_t t = (int64_t)bx * 256 + threadIdx.x;
  if (t >= (int64_t)L * h4) return;
  const int l = (int)(t / h4), c = (int)(t % h4) * 4;
  const int b0 = by * bchunk, b1 = min(B, b0 + bchunk);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int b = b0; b < b1; ++b) s += *reinterpret_cast<const f32x4*>(x + ((int64_t)b * L + l) * H + c);
  *reinterpret_cast<f32x4*>(partial + ((int64_t)by * L + l) * H + c) = s;
}
__global__ __launch_bounds__(256) void batch_colsum_kernel(const float* x, int B, int L, int H, int bchunk, float* partial) {
  batch_colsum_body(x, B, L, H, bchunk, partial, (int)blockIdx.x, (int)blockIdx.y);
}

// Item-table scatter-add in 64-bit FIXED POINT: integer sums do not depend on the order in which the adds arrive, so the item-table
// gradient -- the one sum of a train step that float atomics left order-dependent -- comes out bit for bit the same on every run.
// fix [table_rows * H] and hot [HOT_SLOTS][hot_rows * H] (int64, zeroed by the caller, followed by the poison word) are converted and
// added to the float gradient by the reduce launch that follows (B4rReduceJob::fix).
// Destination rows [0, hot_rows) (PAD / MASK / UNK: [MASK] alone is ~20 % of all tokens) are summed in LDS first and leave the
// workgroup once, into slot (workgroup % HOT_SLOTS) -- every workgroup on one row runs an order of magnitude below the atomic rate.
// Range (round 4): units of 2^-36.  A value of magnitude < 2^18 converts without overflow and a 64-bit sum holds 2^27 in total: with
// the 2^-44 units of round 3 a row's sum wrapped silently at 2^19 = 5.2e5 -- reachable by the [MASK] row of a diverging run, whose
// d(loss_SUM) contributions number in the tens of thousands.  Values of magnitude >= 2^-12 are multiples of the unit (fp32 carries 24
// bits): only smaller contributions are rounded at all, to 7e-12 absolute.  A contribution that is not finite or reaches 2^18 in
// magnitude -- float atomics would have carried an Inf / NaN into the gradient -- sets a sticky POISON word next to the sums; the
// closing reduction then stores NaN for every element of the table gradient, so the step's gradient norm, loss checks and the
// optimizer see the failure instead of finite garbage.
constexpr float FIX_SCALE = 68719476736.f;             // 2^36
constexpr float FIX_UNSCALE = 1.f / 68719476736.f;
constexpr float FIX_LIMIT = 262144.f;                  // 2^18
__device__ __forceinline__ void scatter_fixed_rows_body(const float* src, const int64_t* idx, int n, int H, long long* fix,
                                                        int64_t dst_rows, int hot_rows, long long* hot, const int block,
                                                        const int nblocks, int* poison) {
  extern __shared__ unsigned long long s_hot64[];
  for (int k = threadIdx.x; k < hot_rows * H; k += 256) s_hot64[k] = 0ull;
  if (hot_rows > 0) __syncthreads();
  const int64_t total = (int64_t)n * H;
  for (int64_t t = (int64_t)block * 256 + threadIdx.x; t < total; t += (int64_t)nblocks * 256) {
    const int i = (int)(t / H), c = (int)(t % H);
    const int64_t r = idx[i];
    if (r < 0 || r >= dst_rows) continue;
    const float val = src[(int64_t)i * H + c];
    if (!(fabsf(val) < FIX_LIMIT)) { atomicOr(poison, 1); continue; }   // (also true for NaN) rare: one atomic per offending element
    const unsigned long long q = (unsigned long long)__float2ll_rn(val * FIX_SCALE);
    if (r < hot_rows) atomicAdd(&s_hot64[(int)r * H + c], q);
    else atomicAdd(reinterpret_cast<unsigned long long*>(fix) + r * H + c, q);
  }
  if (hot_rows > 0) {
    __syncthreads();
    for (int k = threadIdx.x; k < hot_rows * H; k += 256) {
      const unsigned long long v = s_hot64[k];
      if (v != 0ull) atomicAdd(reinterpret_cast<unsigned long long*>(hot) + (int64_t)(block % HOT_SLOTS) * hot_rows * H + k, v);
    }
  }
}

// the two gradients of the embedding stage from d(item row + position row) [B*L, H] in ONE launch: the item-table scatter-add
// (workgroups [0, n_scatter)) and the batch sums of the position table (the rest, gx per batch slice): both only read x
// -----------------------------------------------------------------------------------------------------------
// softmax cross entropy + argmax metrics, one workgroup per row
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_ce_kernel(float* logits, int V, int ld, const int64_t* y_true,
                                                         float* row_out, int want_grad) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ float s_sum[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* row = logits + (int64_t)m * ld;
  const int nv = ld / 4;
  int64_t y = y_true[m];
  const bool y_ok = (y >= 0 && y < V);
  const float y_logit = y_ok ? row[y] : 0.f;

  float best = -INFINITY; int bidx = 0x7fffffff;
  for (int v = tid; v < nv; v += 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * v + e;
      if (c < V && (x[e] > best || (x[e] == best && c < bidx))) { best = x[e]; bidx = c; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = bidx; }
  __syncthreads();
  best = s_val[0]; bidx = s_idx[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bidx)) { best = s_val[w]; bidx = s_idx[w]; }

  float sum = 0.f;
  for (int v = tid; v < nv; v += 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * v);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * v + e < V) sum += __expf(x[e] - best);
  }
  sum = b4r_wave_sum(sum);
  if (lane == 0) s_sum[wave] = sum;
  __syncthreads();
  sum = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
  const float lse = best + __logf(sum);
  const bool valid = (y != 0);
  if (tid == 0) {
    row_out[4 * (int64_t)m + 0] = (valid && y_ok) ? (lse - y_logit) : 0.f;
    row_out[4 * (int64_t)m + 1] = valid ? 1.f : 0.f;
    row_out[4 * (int64_t)m + 2] = (valid && (int64_t)bidx == y) ? 1.f : 0.f;
    row_out[4 * (int64_t)m + 3] = ((int64_t)bidx == y) ? 1.f : 0.f;
  }
  if (want_grad) {
    __syncthreads();  // every thread has finished reading the row
    for (int v = tid; v < nv; v += 256) {
      f32x4 x = *reinterpret_cast<const f32x4*>(row + 4 * v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * v + e;
        float g = 0.f;
        if (valid && c < V) g = __expf(x[e] - lse) - ((int64_t)c == y ? 1.f : 0.f);
        x[e] = g;
      }
      *reinterpret_cast<f32x4*>(row + 4 * v) = x;
    }
  }
}

// Ordered sum of the per-row scalars [M][4] (loss term, valid flag, correct-masked, correct-all) by ONE workgroup of 256 threads
// (fixed summation order => bitwise reproducible): thread t adds its rows t, t + 256, ... in order (the loads of 8 rows in
// flight), then threads 0..3 add the 256 partial sums of one component each in thread order.  s: [4][256] floats of LDS.
__device__ __forceinline__ void loss_rows_reduce(const float* rows, int M, float (*s)[256], float (&out)[4]) {
  const int tid = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 16;   // loads in flight per thread: the single workgroup that runs this is latency-bound (8 in flight: 8 us at M = 10240)
  for (int m0 = tid; m0 < M; m0 += U * 256) {
    f32x4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      r[u] = m0 + 256 * u < M ? *reinterpret_cast<const f32x4*>(rows + 4 * (int64_t)(m0 + 256 * u)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (m0 + 256 * u < M) acc += r[u];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q][tid] = acc[q];
  __syncthreads();
  // the 256 partial sums of a component: lane l of wave q adds partials 4l .. 4l+3 in order, then a fixed butterfly over the 64 lanes
  // (every run adds the same numbers in the same order; 256 dependent additions by one thread took 2 us)
  if (tid < 256) {
    const int q = tid >> 6, l = tid & 63;
    float v = ((s[q][4 * l] + s[q][4 * l + 1]) + s[q][4 * l + 2]) + s[q][4 * l + 3];
    v = b4r_wave_sum(v);
    if (l == 0) s[q][0] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = s[q][0];
}

// fin_rows (optional): ONE more workgroup at the end of the grid forms the step's loss / metric sums from the head's per-row scalars
// (what zero2_kernel's last workgroup does when the rows exist before the backward starts; with the head's merge folded into the dE
// launch they only exist from there on) and copies them behind the gradients (tail, optional)
__global__ __launch_bounds__(256) void embed_grads_kernel(const float* x, const int64_t* ids, int n, int H, long long* fix,
                                                          int64_t table_rows, int hot_rows, long long* hot, int n_scatter, int B, int L,
                                                          int bchunk, int gx, float* partial, int n_colsum, const float* fin_rows,
                                                          int fin_M, float* state_f, float* tail) {
  if ((int)blockIdx.x < n_scatter) {
    scatter_fixed_rows_body(x, ids, n, H, fix, table_rows, hot_rows, hot, (int)blockIdx.x, n_scatter,
                            reinterpret_cast<int*>(hot + (int64_t)HOT_SLOTS * hot_rows * H));
  } else if ((int)blockIdx.x < n_scatter + n_colsum) {
    const int k = (int)blockIdx.x - n_scatter;
    batch_colsum_body(x, B, L, H, bchunk, partial, k % gx, k / gx);
  } else {
    __shared__ float s[4][256];
    float r[4];
    loss_rows_reduce(fin_rows, fin_M, s, r);
    if (threadIdx.x == 0) {   // b4r_train_state floats: [4] loss_sum [5] valid_count [6] correct_masked [7] correct_all [8] slots_all [9] [10]
      state_f[4] = r[0]; state_f[5] = r[1]; state_f[6] = r[2]; state_f[7] = r[3];
      state_f[8] = (float)fin_M; state_f[9] = 0.f; state_f[10] = 0.f;
    }
    if (tail != nullptr && threadIdx.x < 8) tail[threadIdx.x] = threadIdx.x < 4 ? r[threadIdx.x] : (threadIdx.x == 4 ? (float)fin_M : 0.f);
  }
}

// the sums into the state: b4r_loss's last launch (the logits-free head's rows or softmax_ce_kernel's)
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* row_out, int M, b4r_train_state* st, int overwrite) {
  __shared__ float s[4][256];
  float r[4];
  loss_rows_reduce(row_out, M, s, r);
  if (threadIdx.x == 0) {
    if (overwrite) {   // what b4r_state_begin_step + accumulation would leave (b4r_train_step saves that launch)
      st->loss_sum = 0.f; st->valid_count = 0.f; st->correct_masked = 0.f; st->correct_all = 0.f; st->slots_all = 0.f;
      st->grad_sqnorm = 0.f; st->grad_norm = 0.f;
    }
    st->loss_sum += r[0];
    st->valid_count += r[1];
    st->correct_masked += r[2];
    st->correct_all += r[3];
    st->slots_all += (float)M;
  }
}

__global__ void state_begin_kernel(b4r_train_state* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->loss_sum = 0.f; st->valid_count = 0.f; st->correct_masked = 0.f; st->correct_all = 0.f; st->slots_all = 0.f;
    st->grad_sqnorm = 0.f; st->grad_norm = 0.f;
  }
}

// -----------------------------------------------------------------------------------------------------------
// global norm + AdamW
// -----------------------------------------------------------------------------------------------------------
// tail (optional): the reduced sums behind the gradients go back into the state before the optimizer kernel reads valid_count
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* g, int64_t n, float* partial, const float* tail = nullptr,
                                                             float* state_f = nullptr) {
  __shared__ float s[4];
  if (tail != nullptr && blockIdx.x == 0 && threadIdx.x < 5) state_f[4 + threadIdx.x] = tail[threadIdx.x];
  const int64_t n4 = n / 4;
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t b = (int64_t)blockIdx.x * per, e = min(n4, b + per);
  float acc = 0.f;
  for (int64_t i = b + threadIdx.x; i < e; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
    acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == gridDim.x - 1) {
    for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
  }
  acc = b4r_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* partial, int np, b4r_train_state* st) {
  __shared__ float s[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) a += partial[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) st->grad_sqnorm = s[0];
}

struct AdamP {
  float* p; const float* g; float* m; float* v;
  int64_t n, n_decay;
  b4r_adamw_config hp;
  b4r_train_state* st;
};

// WarmUp over PolynomialDecay(power 1), float32 like TF: adam_w_optimizer.py:22-36
__device__ __forceinline__ float lr_schedule(const b4r_adamw_config& hp, int64_t step) {
  const float s = (float)step;
  if (hp.num_warmup_steps > 0 && s < (float)hp.num_warmup_steps) return hp.init_lr * (s / (float)hp.num_warmup_steps);
  const float T = (float)hp.num_train_steps;
  const float gs = fminf(s, T);
  const float pr = gs / T;
  return (hp.init_lr - hp.end_lr) * (1.0f - pr) + hp.end_lr;
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamP a) {
  const int64_t step = a.st->step;
  const float cnt = a.st->valid_count;
  const float inv_cnt = cnt > 0.f ? 1.0f / cnt : 1.0f;
  const float gnorm = sqrtf(a.st->grad_sqnorm) * inv_cnt;
  const float clip_scale = a.hp.clip_norm > 0.f ? a.hp.clip_norm / fmaxf(gnorm, a.hp.clip_norm) : 1.0f;
  const float lr_t = lr_schedule(a.hp, step);
  const float t = (float)(step + 1);
  const float b1p = powf(a.hp.beta_1, t), b2p = powf(a.hp.beta_2, t);
  const float alpha = lr_t * sqrtf(1.0f - b2p) / (1.0f - b1p);
  const float omb1 = 1.0f - a.hp.beta_1, omb2 = 1.0f - a.hp.beta_2;
  const float wd = a.hp.weight_decay_rate, eps = a.hp.epsilon;
  const int64_t n4 = a.n / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 p = *reinterpret_cast<f32x4*>(a.p + 4 * i);
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.g + 4 * i);
    f32x4 m = *reinterpret_cast<f32x4*>(a.m + 4 * i);
    f32x4 v = *reinterpret_cast<f32x4*>(a.v + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = (g[e] * inv_cnt) * clip_scale;
      float pe = p[e];
      if (wd != 0.f && (a.hp.decay_mask ? a.hp.decay_mask[4 * i + e] != 0 : 4 * i + e < a.n_decay)) pe -= lr_t * pe * wd;   // _decay_weights_op, before Adam
      m[e] += (ge - m[e]) * omb1;
      v[e] += (ge * ge - v[e]) * omb2;
      pe -= (m[e] * alpha) / (sqrtf(v[e]) + eps);
      p[e] = pe;
    }
    *reinterpret_cast<f32x4*>(a.p + 4 * i) = p;
    *reinterpret_cast<f32x4*>(a.m + 4 * i) = m;
    *reinterpret_cast<f32x4*>(a.v + 4 * i) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { a.st->grad_norm = gnorm; a.st->lr = lr_t; }
}

// clip + decay + Adam + step advance in ONE launch after sqnorm_partial_kernel (b4r_optimizer_step): every workgroup sums the
// np partial norms itself (same order, same value everywhere), the last one to finish (ticket) publishes norm / lr and
// advances the step -- the other workgroups have read the state by then.
__global__ __launch_bounds__(256) void adamw_fused_kernel(AdamP a, const float* partial, int np, unsigned int* ticket) {
  __shared__ float s_red[256];
  // the operands of this thread's first turn (normally its only one) and the state are requested before the norm is summed: the
  // barriers of that sum would otherwise stand between their latency and their use
  const int64_t n4 = a.n / 4;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i0c = i0 < n4 ? i0 : 0;
  f32x4 p0 = *reinterpret_cast<f32x4*>(a.p + 4 * i0c);
  const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.g + 4 * i0c);
  f32x4 m0 = *reinterpret_cast<f32x4*>(a.m + 4 * i0c);
  f32x4 v0 = *reinterpret_cast<f32x4*>(a.v + 4 * i0c);
  const int64_t step = a.st->step;
  const float cnt = a.st->valid_count;
  float acc = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
  s_red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
    __syncthreads();
  }
  const float sqnorm = s_red[0];
  const float inv_cnt = cnt > 0.f ? 1.0f / cnt : 1.0f;
  const float gnorm = sqrtf(sqnorm) * inv_cnt;
  const float clip_scale = a.hp.clip_norm > 0.f ? a.hp.clip_norm / fmaxf(gnorm, a.hp.clip_norm) : 1.0f;
  const float lr_t = lr_schedule(a.hp, step);
  const float t = (float)(step + 1);
  const float b1p = powf(a.hp.beta_1, t), b2p = powf(a.hp.beta_2, t);
  const float alpha = lr_t * sqrtf(1.0f - b2p) / (1.0f - b1p);
  const float omb1 = 1.0f - a.hp.beta_1, omb2 = 1.0f - a.hp.beta_2;
  const float wd = a.hp.weight_decay_rate, eps = a.hp.epsilon;
  for (int64_t i = i0; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 p, g, m, v;
    if (i == i0) { p = p0; g = g0; m = m0; v = v0; }
    else {
      p = *reinterpret_cast<f32x4*>(a.p + 4 * i);
      g = *reinterpret_cast<const f32x4*>(a.g + 4 * i);
      m = *reinterpret_cast<f32x4*>(a.m + 4 * i);
      v = *reinterpret_cast<f32x4*>(a.v + 4 * i);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = (g[e] * inv_cnt) * clip_scale;
      float pe = p[e];
      if (wd != 0.f && (a.hp.decay_mask ? a.hp.decay_mask[4 * i + e] != 0 : 4 * i + e < a.n_decay)) pe -= lr_t * pe * wd;   // _decay_weights_op, before Adam
      m[e] += (ge - m[e]) * omb1;
      v[e] += (ge * ge - v[e]) * omb2;
      pe -= (m[e] * alpha) / (sqrtf(v[e]) + eps);
      p[e] = pe;
    }
    *reinterpret_cast<f32x4*>(a.p + 4 * i) = p;
    *reinterpret_cast<f32x4*>(a.m + 4 * i) = m;
    *reinterpret_cast<f32x4*>(a.v + 4 * i) = v;
  }
  // every thread has USED the state values it loaded by now; the barrier only has to order those uses before the ticket, it need
  // not wait for this workgroup's stores (s_barrier alone: __syncthreads would add s_waitcnt vmcnt(0))
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) {
    // no release fence: the other workgroups only READ the state, and those loads have completed (their values were used
    // above) before this relaxed device-scope atomic is issued; a fence here would write back each XCD's dirty L2 lines
    // (the parameters just updated) once per workgroup -- measured 12 us
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      a.st->grad_sqnorm = sqnorm; a.st->grad_norm = gnorm; a.st->lr = lr_t;
      a.st->step = step + 1; a.st->step_lo = (uint32_t)(step + 1);
      *ticket = 0u;   // ready for the next step
    }
  }
}

__global__ void state_advance_kernel(b4r_train_state* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { st->step += 1; st->step_lo = (uint32_t)st->step; }
}

}  // namespace

// ===============================================================================================================
// C ABI
// ===============================================================================================================
extern "C" int b4r_embed_ln_fwd(const int64_t* ids, int32_t B, int32_t L, const float* table, int32_t V,
                                const float* pos_table, const float* gamma, const float* beta, int32_t H, float eps,
                                float* out, float* mean, float* rstd, const uint32_t* rng, float dropout,
                                b4r_stream_t stream) {
  B4R_CHECK_ARG(ids && table && pos_table && gamma && beta && out, B4R_E_BADARG, "b4r_embed_ln_fwd: null argument");
  B4R_CHECK_ARG(B > 0 && L > 0 && V > 0, B4R_E_SHAPE, "b4r_embed_ln_fwd: bad shape");
  LnFwdP p{};
  p.ids = ids; p.table = table; p.pos_table = pos_table; p.L = L; p.V = V;
  p.gamma = gamma; p.beta = beta; p.y = out; p.mean = mean; p.rstd = rstd;
  p.rows = B * L; p.H = H; p.eps = eps;
  p.drop = b4r_make_drop(rng, B4R_STREAM_EMB, dropout, 1);
  int rc = launch_ln_fwd<true>(p, (hipStream_t)stream);
  if (rc) return rc;
  B4R_CHECK_LAUNCH("b4r_embed_ln_fwd");
  return B4R_OK;
}

extern "C" int b4r_ln_fwd(const float* z, int32_t rows, int32_t H, const float* gamma, const float* beta, float eps,
                          float* y, float* mean, float* rstd, b4r_stream_t stream) {
  B4R_CHECK_ARG(z && gamma && beta && y, B4R_E_BADARG, "b4r_ln_fwd: null argument");
  B4R_CHECK_ARG(rows > 0, B4R_E_SHAPE, "b4r_ln_fwd: bad shape");
  LnFwdP p{};
  p.z = z; p.gamma = gamma; p.beta = beta; p.y = y; p.mean = mean; p.rstd = rstd;
  p.rows = rows; p.H = H; p.eps = eps; p.L = 1; p.V = 1;
  p.drop = b4r_make_drop(nullptr, 0, 0.f, 0);
  int rc = launch_ln_fwd<false>(p, (hipStream_t)stream);
  if (rc) return rc;
  B4R_CHECK_LAUNCH("b4r_ln_fwd");
  return B4R_OK;
}

extern "C" int64_t b4r_ln_bwd_scratch_floats(int32_t rows, int32_t H) {
  if (rows <= 0 || H < 32) return 0;
  return (int64_t)ln_bwd_grid(rows, H) * 2 * H;
}

namespace {
// dgamma[c] = sum_s partial[s][c], dbeta[c] = sum_s partial[s][H + c]; 4 columns x 64 slab lanes per workgroup,
// combined in a fixed order
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* partial, int S, int H, float* dgamma, float* dbeta) {
  __shared__ float sp[64][4];
  const int cl = threadIdx.x & 3, zl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  float s = 0.f;
  if (c < 2 * H) {
    for (int z = zl; z < S; z += 64) s += partial[(int64_t)z * 2 * H + c];
  }
  sp[zl][cl] = s;
  __syncthreads();
  if (zl == 0 && c < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int z = 0; z < 64; ++z) t += sp[z][cl];
    if (c < H) dgamma[c] = t; else dbeta[c - H] = t;
  }
}
}  // namespace

int b4r_ln_bwd_launch(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                      int rows, int H, float* dz, float* dgamma, float* dbeta, float* scratch, const int64_t* ids,
                      const float* table, const float* pos_table, int L, int V, DropArgs drop, hipStream_t stream,
                      const float* gelu_pre, const B4rHeadMerge* merge) {
  LnBwdP p{};
  p.gelu_pre = gelu_pre;
  if (merge != nullptr)
    p.merge = HeadMergeP{merge->part, merge->slices, merge->M, merge->V, merge->T, merge->E, merge->bias, merge->y, merge->row_out,
                         merge->lse_out, merge->ylab};
  p.dy = dy; p.z = z; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.dz = dz; p.partial = scratch;
  p.ids = ids; p.table = table; p.pos_table = pos_table; p.L = L; p.V = V;
  p.rows = rows; p.H = H; p.drop = drop;
  const int grid = ln_bwd_grid(rows, H);
  int rc = ids ? launch_ln_bwd<true>(p, grid, stream) : launch_ln_bwd<false>(p, grid, stream);
  if (rc) return rc;
  B4R_CHECK_LAUNCH("ln_bwd");
  if (dbeta == dgamma + H) {   // gamma and beta adjacent (always so in the flat gradient buffer): one [1, 2H] strip
    B4rReduceJob job{scratch, nullptr, nullptr, dgamma, nullptr, nullptr, grid, 1, 2 * H, 2 * H, 0};
    if (b4r_reduce_queue_push(job)) return B4R_OK;
  }
  hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3(b4r_cdiv(2 * H, 4)), dim3(256), 0, stream, scratch, grid, H,
                     dgamma, dbeta);
  B4R_CHECK_LAUNCH("ln_bwd reduce");
  return B4R_OK;
}

extern "C" int b4r_ln_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                          int32_t rows, int32_t H, float* dz, float* dgamma, float* dbeta, float* scratch,
                          b4r_stream_t stream) {
  B4R_CHECK_ARG(dy && z && mean && rstd && gamma && dz && dgamma && dbeta && scratch, B4R_E_BADARG,
                "b4r_ln_bwd: null argument");
  B4R_CHECK_ARG(rows > 0, B4R_E_SHAPE, "b4r_ln_bwd: bad shape");
  return b4r_ln_bwd_launch(dy, z, mean, rstd, gamma, rows, H, dz, dgamma, dbeta, scratch, nullptr, nullptr, nullptr, 1,
                           1, b4r_make_drop(nullptr, 0, 0.f, 0), (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int b4r_gather_rows(const float* src, int32_t src_ld, const int64_t* idx, int64_t idx_add_per, int32_t per,
                               int32_t n, int32_t H, float* dst, b4r_stream_t stream) {
  B4R_CHECK_ARG(src && idx && dst, B4R_E_BADARG, "b4r_gather_rows: null argument");
  B4R_CHECK_ARG(n > 0 && H > 0 && H % 4 == 0 && per > 0 && src_ld % 4 == 0, B4R_E_SHAPE, "b4r_gather_rows: bad shape");
  int grid = b4r_cdiv((int64_t)n * (H / 4), 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, src_ld, idx, idx_add_per,
                     per, n, H, dst);
  B4R_CHECK_LAUNCH("b4r_gather_rows");
  return B4R_OK;
}

// hot_scratch (optional, with hot_rows > 0): b4r_scatter_hot_scratch_floats(hot_rows, H) floats, ZEROED by the caller
int64_t b4r_scatter_hot_scratch_floats(int hot_rows, int H) { return (int64_t)HOT_SLOTS * hot_rows * H; }

int b4r_scatter_add_rows_impl(const float* src, const int64_t* idx, int64_t idx_add_per, int per, int n, int H,
                              float* dst, int dst_ld, const int64_t* skip_if_zero, int64_t dst_rows, int hot_rows,
                              float* hot_scratch, hipStream_t stream) {
  int grid = b4r_cdiv((int64_t)n * H, 256);
  const int cap = hot_rows > 0 ? 1024 : 8192;
  if (grid > cap) grid = cap;
  if (hot_rows <= 0) hot_scratch = nullptr;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(256), (size_t)hot_rows * H * sizeof(float), stream, src, idx,
                     idx_add_per, per, n, H, dst, dst_ld, skip_if_zero, dst_rows, hot_rows, hot_scratch);
  B4R_CHECK_LAUNCH("b4r_scatter_add_rows");
  if (hot_scratch)   // joins the caller's reduce queue when one is active
    return b4r_launch_slab_reduce_full(hot_scratch, HOT_SLOTS, hot_rows, H, dst, dst_ld, 1, nullptr, nullptr, nullptr, nullptr, stream);
  return B4R_OK;
}

extern "C" int b4r_scatter_add_rows(const float* src, const int64_t* idx, int64_t idx_add_per, int32_t per, int32_t n,
                                    int32_t H, float* dst, int32_t dst_ld, const int64_t* skip_if_zero,
                                    b4r_stream_t stream) {
  B4R_CHECK_ARG(src && idx && dst, B4R_E_BADARG, "b4r_scatter_add_rows: null argument");
  B4R_CHECK_ARG(n > 0 && H > 0 && H % 4 == 0 && per > 0 && dst_ld % 4 == 0, B4R_E_SHAPE, "b4r_scatter_add_rows: bad shape");
  return b4r_scatter_add_rows_impl(src, idx, idx_add_per, per, n, H, dst, dst_ld, skip_if_zero, (int64_t)1 << 62, 0, nullptr,
                                   (hipStream_t)stream);
}

// dpos[l][c] = sum_b x[(b*L+l)*H + c]; scratch >= ceil(B/16)*L*H floats
int b4r_batch_colsum(const float* x, int B, int L, int H, float* dpos, float* scratch, hipStream_t stream) {
  const int bchunk = 16;
  const int S = b4r_cdiv(B, bchunk);
  dim3 grid(b4r_cdiv((int64_t)L * (H / 4), 256), S);
  hipLaunchKernelGGL(batch_colsum_kernel, grid, dim3(256), 0, stream, x, B, L, H, bchunk, scratch);
  B4R_CHECK_LAUNCH("batch_colsum");
  return b4r_launch_slab_reduce_full(scratch, S, L, H, dpos, H, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

// item-table and position-table gradients of the embedding stage in one launch (b4r_backward's tail).  fixed: b4r_embed_fixed_floats
// floats, ZEROED by the caller (the 64-bit fixed-point sums of scatter_fixed_rows_body); colsum_scratch >= ceil(B/16)*L*H floats.
// Both results are completed by reduce jobs: inside the caller's queue when one is active (the item table's job is the one that
// already sums the head's slabs into table_grad, if there is one: table_grad = slabs + fixed in ONE pass), else launched here.
// (+ 4 floats: the sticky poison word behind the sums, zeroed with them)
int64_t b4r_embed_fixed_floats(int64_t V, int H, int hot_rows) { return 2 * (V * H + (int64_t)HOT_SLOTS * hot_rows * H) + 4; }
int b4r_embed_grads(const float* x, const int64_t* ids, int B, int L, int H, float* table_grad, int64_t V, int hot_rows,
                    float* fixed, float* dpos, float* colsum_scratch, hipStream_t stream, const float* fin_rows, int fin_M,
                    b4r_train_state* state, float* tail) {
  const int n = B * L, bchunk = 16, S = b4r_cdiv(B, bchunk);
  int n_scatter = b4r_cdiv((int64_t)n * H, 256);
  if (n_scatter > 1024) n_scatter = 1024;
  if (hot_rows < 0) hot_rows = 0;
  long long* fix = reinterpret_cast<long long*>(fixed);
  long long* hot = fix + V * H;
  const int gx = b4r_cdiv((int64_t)L * (H / 4), 256);
  hipLaunchKernelGGL(embed_grads_kernel, dim3(n_scatter + gx * S + (fin_rows ? 1 : 0)), dim3(256), (size_t)hot_rows * H * sizeof(long long),
                     stream, x, ids, n, H, fix, V, hot_rows, hot, n_scatter, B, L, bchunk, gx, colsum_scratch, gx * S, fin_rows, fin_M,
                     reinterpret_cast<float*>(state), tail);
  B4R_CHECK_LAUNCH("embedding gradients (scatter-add + position sums)");
  const int* poison = reinterpret_cast<const int*>(hot + (int64_t)HOT_SLOTS * hot_rows * H);
  if (!b4r_reduce_queue_attach_fixed(table_grad, fix, hot, hot_rows * H, HOT_SLOTS, poison)) {
    B4rReduceJob job{nullptr, nullptr, nullptr, table_grad, nullptr, nullptr, 0, (int)V, H, H, 1};
    job.fix = fix; job.fix_hot = hot; job.fix_hot_elems = hot_rows * H; job.fix_slots = HOT_SLOTS; job.fix_poison = poison;
    int rc = b4r_launch_reduce_job(job, stream);
    if (rc) return rc;
  }
  return b4r_launch_slab_reduce_full(colsum_scratch, S, L, H, dpos, H, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int b4r_softmax_ce(float* logits, int32_t M, int32_t V, int32_t ld, const int64_t* y_true,
                              float* row_scratch, b4r_train_state* state, int32_t want_grad, b4r_stream_t stream) {
  B4R_CHECK_ARG(logits && y_true && row_scratch && state, B4R_E_BADARG, "b4r_softmax_ce: null argument");
  B4R_CHECK_ARG(M > 0 && V > 0 && ld >= V, B4R_E_SHAPE, "b4r_softmax_ce: bad shape");
  B4R_CHECK_ARG(ld % 4 == 0 && b4r_aligned16(logits), B4R_E_ALIGN, "b4r_softmax_ce: logits need ld %% 4 == 0 and 16-byte alignment");
  hipLaunchKernelGGL(softmax_ce_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, logits, V, ld, y_true, row_scratch,
                     want_grad & 1);
  B4R_CHECK_LAUNCH("b4r_softmax_ce");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_scratch, M, state, (want_grad >> 2) & 1);
  B4R_CHECK_LAUNCH("b4r_softmax_ce finalize");
  return B4R_OK;
}

// zero two float regions (16-byte aligned, sizes multiples of 4 floats) in one launch.  Used instead of hipMemsetAsync by
// the backward pass: one launch instead of two, and an ordinary kernel node when the step is captured into a hipGraph
// (memset nodes of a replayed graph were observed to leave the regions untouched from the second replay on).
// tail (optional): the 8 floats behind the gradient buffer receive the step's five sums from the state (loss_sum, valid_count,
// correct_masked, correct_all, slots_all), so that ONE all-reduce of [gradients | tail] carries them
// fin_rows (optional): the LAST workgroup also does what ce_finalize_kernel does in overwrite mode -- the ordered sum of the fused
// head's per-row scalars into the state (loss_rows_reduce: the same summation order, bit for bit) -- before the tail copy, so that
// b4r_backward needs no b4r_loss launch in front of it (B4R_FLAG_LOSS_SUMS)
// rider (rider_blocks > 0): the last rider_blocks workgroups form tile records of the 32 x 32-tile masked-LM head instead (one each)
__global__ __launch_bounds__(256) void zero2_kernel(float* a, int64_t na4, float* b, int64_t nb4, float* tail, float* state_f,
                                                    const float* fin_rows, int fin_M, H32PackP rider, int rider_blocks) {
  __shared__ float s[4][256];
  const int main_blocks = gridDim.x - rider_blocks;
  if ((int)blockIdx.x >= main_blocks) {
    h32_pack_tile_any(rider, (int)blockIdx.x - main_blocks);
    return;
  }
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const bool fin = fin_rows != nullptr && (int)blockIdx.x == main_blocks - 1;
  if (fin) {
    float r[4];
    loss_rows_reduce(fin_rows, fin_M, s, r);
    if (threadIdx.x == 0) {   // b4r_train_state floats: [4] loss_sum [5] valid_count [6] correct_masked [7] correct_all [8] slots_all [9] [10]
      state_f[4] = r[0]; state_f[5] = r[1]; state_f[6] = r[2]; state_f[7] = r[3];
      state_f[8] = (float)fin_M; state_f[9] = 0.f; state_f[10] = 0.f;
    }
    __syncthreads();
  }
  if (tail != nullptr && (fin_rows != nullptr ? fin : blockIdx.x == 0) && threadIdx.x < 8)
    tail[threadIdx.x] = threadIdx.x < 5 ? state_f[4 + threadIdx.x] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na4 + nb4; i += (int64_t)main_blocks * 256) {
    if (i < na4) *reinterpret_cast<f32x4*>(a + 4 * i) = z;
    else *reinterpret_cast<f32x4*>(b + 4 * (i - na4)) = z;
  }
}
// rider / rider_blocks: an H32PackP job (b4r_head_rx_dE_pack_job) carried by rider_blocks extra workgroups of the launch
int b4r_zero2(float* a, int64_t na, float* b, int64_t nb, hipStream_t stream, float* tail, b4r_train_state* state,
              const float* fin_rows, int fin_M, const void* rider, int rider_blocks) {
  B4R_CHECK_ARG(na % 4 == 0 && nb % 4 == 0 && b4r_aligned16(a) && b4r_aligned16(b), B4R_E_ALIGN, "zero2: regions must be 16-byte granular");
  B4R_CHECK_ARG(fin_rows == nullptr || (state != nullptr && fin_M > 0), B4R_E_BADARG, "zero2: the loss sums need the state");
  int64_t n4 = (na + nb) / 4;
  int grid = (int)((n4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  H32PackP job{};
  if (rider != nullptr && rider_blocks > 0) job = *reinterpret_cast<const H32PackP*>(rider);
  else rider_blocks = 0;
  hipLaunchKernelGGL(zero2_kernel, dim3(grid + rider_blocks), dim3(256), 0, stream, a, na / 4, b, nb / 4, tail, reinterpret_cast<float*>(state),
                     fin_rows, fin_M, job, rider_blocks);
  B4R_CHECK_LAUNCH(fin_rows ? "zero fill + loss sums" : "zero fill");
  return B4R_OK;
}

int b4r_ce_finalize_launch(const float* row_scratch, int M, b4r_train_state* state, int overwrite, hipStream_t stream) {
  B4R_CHECK_ARG(row_scratch && state && M > 0, B4R_E_BADARG, "ce_finalize: bad argument");
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, stream, row_scratch, M, state, overwrite);
  B4R_CHECK_LAUNCH("b4r_loss finalize");
  return B4R_OK;
}

extern "C" int b4r_state_begin_step(b4r_train_state* state, b4r_stream_t stream) {
  B4R_CHECK_ARG(state, B4R_E_BADARG, "b4r_state_begin_step: null state");
  hipLaunchKernelGGL(state_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
  B4R_CHECK_LAUNCH("b4r_state_begin_step");
  return B4R_OK;
}

extern "C" int b4r_global_sqnorm(const float* g, int64_t n, float* scratch, b4r_train_state* state, b4r_stream_t stream) {
  B4R_CHECK_ARG(g && scratch && state, B4R_E_BADARG, "b4r_global_sqnorm: null argument");
  B4R_CHECK_ARG(n > 0, B4R_E_SHAPE, "b4r_global_sqnorm: bad size");
  B4R_CHECK_ARG(b4r_aligned16(g), B4R_E_ALIGN, "b4r_global_sqnorm: buffer must be 16-byte aligned");
  int np = (int)((n / 4 + 1023) / 1024);
  if (np > 1024) np = 1024;
  if (np < 1) np = 1;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(np), dim3(256), 0, (hipStream_t)stream, g, n, scratch, (const float*)nullptr,
                     (float*)nullptr);
  B4R_CHECK_LAUNCH("b4r_global_sqnorm");
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, np, state);
  B4R_CHECK_LAUNCH("b4r_global_sqnorm final");
  return B4R_OK;
}

// model-level optimizer step: global norm partials, then everything else in one launch.  scratch: >= 1024 floats.  The
// ticket is reserved[0] of the state (zero-initialised by the caller like the rest of the state, reset by the kernel).
// np_given > 0: scratch already holds np_given partial sums of squares of the gradients (b4r_backward's closing reduce launch formed
// them while it wrote the gradients): no norm launch
int b4r_optimizer_fused(const b4r_adamw_config* hp, float* params, const float* grads, float* adam_m, float* adam_v, int64_t n,
                        int64_t n_decay, float* scratch, b4r_train_state* state, hipStream_t stream, int sums_from_tail, int np_given) {
  int np = (int)((n / 4 + 1023) / 1024);
  if (np > 1024) np = 1024;
  if (np < 1) np = 1;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(&state->reserved[0]);
  if (np_given > 0 && !sums_from_tail) {
    np = np_given;
  } else {
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(np), dim3(256), 0, stream, grads, n, scratch,
                       sums_from_tail ? grads + n : (const float*)nullptr, reinterpret_cast<float*>(state));
    B4R_CHECK_LAUNCH("global norm");
  }
  AdamP a;
  a.p = params; a.g = grads; a.m = adam_m; a.v = adam_v; a.n = n; a.n_decay = n_decay; a.hp = *hp; a.st = state;
  int grid = (int)((n / 4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adamw_fused_kernel, dim3(grid), dim3(256), 0, stream, a, (const float*)scratch, np, ticket);
  B4R_CHECK_LAUNCH("optimizer step");
  return B4R_OK;
}

extern "C" int b4r_adamw_step(const b4r_adamw_config* hp, float* params, const float* grads, float* adam_m,
                              float* adam_v, int64_t n, int64_t n_decay, b4r_train_state* state, b4r_stream_t stream) {
  B4R_CHECK_ARG(hp && params && grads && adam_m && adam_v && state, B4R_E_BADARG, "b4r_adamw_step: null argument");
  B4R_CHECK_ARG(n > 0 && n % 4 == 0 && n_decay >= 0 && n_decay <= n, B4R_E_SHAPE, "b4r_adamw_step: n must be a positive multiple of 4");
  B4R_CHECK_ARG(b4r_aligned16(params) && b4r_aligned16(grads) && b4r_aligned16(adam_m) && b4r_aligned16(adam_v),
                B4R_E_ALIGN, "b4r_adamw_step: buffers must be 16-byte aligned");
  AdamP a;
  a.p = params; a.g = grads; a.m = adam_m; a.v = adam_v; a.n = n; a.n_decay = n_decay; a.hp = *hp; a.st = state;
  int grid = (int)((n / 4 + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  B4R_CHECK_LAUNCH("b4r_adamw_step");
  hipLaunchKernelGGL(state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
  B4R_CHECK_LAUNCH("b4r_adamw_step advance");
  return B4R_OK;
}

// -----------------------------------------------------------------------------------------------------------
// batch construction: the masked-LM task of the preprocessor for a whole batch, one wave per sequence
// -----------------------------------------------------------------------------------------------------------
// tokens [B,L]: ids right-padded with 0.  Restates apply_dynamic_masking_task (dataloader_utils.py:186-261) with a
// counter-hash stream instead of python's random: n = tokens that are neither PAD (0) nor UNK (2);
// num = min(P, max(1, int(n * rate))) positions are a uniform subset of [0, n) (random keys, the num smallest win), visited
// in ascending order; each becomes [MASK] (1) with probability mask_rate, a uniform id of the vocabulary minus {PAD, UNK}
// with probability random_rate, else stays.  finetune != 0 restates mask_last_token_only (:264-269).
// row_index (optional): output row r is dataset row row_index[r] of `tokens` (the whole [U, L] matrix of a dataset stays in HBM, a
// batch is an index list); the random stream is keyed by the DATASET row, so a row's mask does not depend on its batch slot.
// row_finetune (optional, per dataset row): != 0 -> last-token mask for that row (the reference mixes both kinds in one shuffled
// training set, bert4rec_dataloader.py:100-108).
__global__ __launch_bounds__(64) void mask_batch_kernel(const int64_t* tokens, const int64_t* row_index, const int64_t* row_finetune,
                                                        int L, int P, int V, double rate, float mask_rate,
                                                        float random_rate, int finetune_all, uint32_t seed_lo, uint32_t seed_hi,
                                                        int64_t* ids_out, int64_t* mask_out, int64_t* labels_out,
                                                        int64_t* pos_out, int64_t* mids_out, int64_t* w_out) {
  __shared__ uint32_t s_key[256];
  __shared__ int s_sel[256];
  const int row = blockIdx.x, lane = threadIdx.x;
  const int64_t srow = row_index ? row_index[row] : (int64_t)row;
  const int finetune = (finetune_all != 0 || (row_finetune != nullptr && row_finetune[srow] != 0)) ? 1 : 0;
  const int64_t* src = tokens + srow * L;
  int len = 0, n = 0;
  for (int i = lane; i < L; i += 64) {
    const int64_t t = src[i];
    len += (t != 0) ? 1 : 0;
    n += (t != 0 && t != 2) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { len += __shfl_xor(len, o, 64); n += __shfl_xor(n, o, 64); }
  const uint32_t rk = b4r_hash32((uint32_t)srow * 0x9E3779B9u + seed_hi);
  int num;
  if (finetune) num = len > 0 ? 1 : 0;
  else num = n > 0 ? min(P, max(1, (int)((double)n * rate))) : 0;
  for (int i = lane; i < L; i += 64) {
    s_key[i] = b4r_hash32(b4r_hash32((uint32_t)i ^ seed_lo) + rk);
    s_sel[i] = 0;
  }
  __syncthreads();
  for (int i = lane; i < L; i += 64) {
    int sel = 0;
    if (finetune) sel = (i == len - 1) ? 1 : 0;
    else if (i < n) {
      int rank = 0;
      const uint32_t ki = s_key[i];
      for (int j = 0; j < n; ++j) { const uint32_t kj = s_key[j]; rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
      sel = rank < num ? 1 : 0;
    }
    s_sel[i] = sel;
  }
  __syncthreads();
  for (int i = lane; i < L; i += 64) {
    const int64_t t = src[i];
    int64_t out = t;
    if (s_sel[i]) {
      int slot = 0;
      for (int j = 0; j < i; ++j) slot += s_sel[j];
      pos_out[(int64_t)row * P + slot] = i;
      mids_out[(int64_t)row * P + slot] = t;
      w_out[(int64_t)row * P + slot] = 1;
      if (finetune) out = 1;
      else {
        const uint32_t h = b4r_hash32(s_key[i] ^ 0x68E31DA4u);
        const float rn = b4r_uniform23(h);   // in (0, 1): with mask_rate = 1 EVERY selected position becomes [MASK] (no label leak)
        if (rn < mask_rate) out = 1;
        else if (rn < mask_rate + random_rate) {
          const uint32_t k = b4r_hash32(h + 0x9E3779B9u) % (uint32_t)(V - 2);   // selectable vocab: every id but PAD, UNK
          out = (k == 0) ? 1 : (int64_t)k + 2;
        }
      }
    }
    ids_out[(int64_t)row * L + i] = out;
    mask_out[(int64_t)row * L + i] = (i < len) ? 1 : 0;
    labels_out[(int64_t)row * L + i] = t;
  }
  for (int s = num + lane; s < P; s += 64) {
    pos_out[(int64_t)row * P + s] = 0; mids_out[(int64_t)row * P + s] = 0; w_out[(int64_t)row * P + s] = 0;
  }
}

extern "C" int b4r_mask_batch(const int64_t* tokens, const int64_t* row_index, const int64_t* row_finetune, int32_t B, int32_t L,
                              int32_t P, int32_t V, double selection_rate,
                              float mask_token_rate, float random_token_rate, int32_t finetune, uint64_t seed,
                              int64_t* input_word_ids, int64_t* input_mask, int64_t* labels, int64_t* masked_lm_positions,
                              int64_t* masked_lm_ids, int64_t* masked_lm_weights, b4r_stream_t stream) {
  B4R_CHECK_ARG(tokens && input_word_ids && input_mask && labels && masked_lm_positions && masked_lm_ids && masked_lm_weights,
                B4R_E_BADARG, "b4r_mask_batch: null argument");
  B4R_CHECK_ARG(B > 0 && L > 0 && L <= 256 && P > 0 && V > 3, B4R_E_SHAPE, "b4r_mask_batch: bad shape (L <= 256)");
  hipLaunchKernelGGL(mask_batch_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, tokens, row_index, row_finetune, L, P, V, selection_rate,
                     mask_token_rate, random_token_rate, finetune, (uint32_t)seed, (uint32_t)(seed >> 32), input_word_ids,
                     input_mask, labels, masked_lm_positions, masked_lm_ids, masked_lm_weights);
  B4R_CHECK_LAUNCH("b4r_mask_batch");
  return B4R_OK;
}

extern "C" float b4r_uniform_from_hash(uint32_t hash_word) { return b4r_uniform23(hash_word); }

// -----------------------------------------------------------------------------------------------------------
// the rows the masked-LM head reads (b4r_mlm_rows): one entry per masked-LM slot
// -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlm_rows_kernel(const int64_t* pos, const int64_t* ids, int L, int P, int M, int* rows, int* n_rows,
                                                       int* row_slot) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m == 0) *n_rows = M;
  if (m >= M) return;
  const int64_t q = pos[m];
  rows[m] = (m / P) * L + (q < 0 ? 0 : (q >= L ? L - 1 : (int)q));   // out-of-range positions are clamped like b4r_gather_rows
  row_slot[m] = ids[m] != 0 ? m : -1;
}

extern "C" int b4r_mlm_rows(const int64_t* masked_lm_positions, const int64_t* masked_lm_ids, int32_t B, int32_t L, int32_t P, int32_t* rows,
                            int32_t* n_rows, int32_t* row_slot, b4r_stream_t stream) {
  B4R_CHECK_ARG(masked_lm_positions && masked_lm_ids && rows && n_rows && row_slot, B4R_E_BADARG, "b4r_mlm_rows: null argument");
  B4R_CHECK_ARG(B > 0 && L > 0 && P > 0 && (int64_t)B * L < (1ll << 31) && (int64_t)B * P < (1ll << 31), B4R_E_SHAPE,
                "b4r_mlm_rows: bad shape (B=%d L=%d P=%d)", B, L, P);
  const int M = B * P;
  hipLaunchKernelGGL(mlm_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, masked_lm_positions, masked_lm_ids, L, P, M,
                     rows, n_rows, row_slot);
  B4R_CHECK_LAUNCH("b4r_mlm_rows");
  return B4R_OK;
}

// -----------------------------------------------------------------------------------------------------------
// negative sampling for the evaluator: one workgroup per ranked slot
// -----------------------------------------------------------------------------------------------------------
// Weighted sampling without replacement (what np.random.choice(vocab, size, False, p) followed by dropping the excluded
// items draws: successive picks proportional to p among what is left) as Gumbel top-k: key_v = log p_v + G_v with
// G_v = -log(-log u_v); the C largest keys, in descending order, are the sample in draw order.  u_v comes from a counter
// hash of (seed, row, v) with 23 bits, so it lies strictly inside (0, 1).
// Round 3: the C draws are no longer made one after the other (a 256-thread tournament + barrier per draw: 127-158 us per 256 rows).
// The sample is the set of the C largest keys, so the row's workgroup SELECTS them: a 4-pass radix select (8 bits per pass, LDS
// histogram) finds the C-th largest key, the keys above it (and the lowest-indexed keys equal to it) are gathered, and their order
// -- descending key, ties by ascending index = the order of the draws -- comes from counting, C x C comparisons spread over the
// workgroup.  Same keys, same tie rule: the output is what the tournament produced.
__device__ __forceinline__ uint32_t key_image(float k) {   // ascending unsigned image of a float (-inf smallest)
  const uint32_t b = __builtin_bit_cast(uint32_t, k);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
constexpr int SAMPLE_MAX_C = 1024;   // draws per row the gather / ordering buffers hold

__global__ __launch_bounds__(256) void sample_candidates_kernel(const float* logp, int V, const int64_t* exclude, int E,
                                                                const int64_t* gt, int C, uint32_t seed_lo,
                                                                uint32_t seed_hi, int64_t* cand, uint8_t* short_flag) {
  extern __shared__ float s_key[];            // [V] keys
  __shared__ int s_hist[256];
  __shared__ uint32_t s_selu[SAMPLE_MAX_C];
  __shared__ int s_seli[SAMPLE_MAX_C];
  __shared__ int s_tie[256];                  // indices of keys equal to the threshold (more than 256 ties: the lowest 256 seen)
  __shared__ int s_cnt[4];                    // [0] finite keys, [1] gathered above the threshold, [2] ties seen, [3] scratch
  __shared__ uint32_t s_prefix;
  __shared__ int s_remaining;
  const int row = blockIdx.x, tid = threadIdx.x;
  const uint32_t rk = b4r_hash32((uint32_t)row * 0x9E3779B9u + seed_hi);
  for (int v = tid; v < V; v += 256) {
    uint32_t h = b4r_hash32((uint32_t)v ^ seed_lo);
    h = b4r_hash32(h + rk);
    const float u = b4r_uniform23(h);   // strictly inside (0, 1)
    s_key[v] = logp[v] - __logf(-__logf(u));  // -inf + finite = -inf: zero-probability items are never drawn
  }
  if (tid < 4) s_cnt[tid] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += 256) {
    const int64_t id = exclude[(int64_t)row * E + e];
    if (id >= 0 && id < V) s_key[id] = -INFINITY;
  }
  const int64_t g = gt ? gt[row] : -1;
  if (tid == 0 && g >= 0 && g < V) s_key[g] = -INFINITY;
  __syncthreads();
  {   // how many keys can be drawn at all
    int n = 0;
    for (int v = tid; v < V; v += 256) n += s_key[v] > -INFINITY ? 1 : 0;
    n = (int)b4r_wave_sum((float)n);
    if ((tid & 63) == 0) atomicAdd(&s_cnt[0], n);
  }
  __syncthreads();
  const int Ce = min(C, s_cnt[0]);            // draws that exist; the rest of the row is -1
  int64_t* out = cand + (int64_t)row * (C + 1);
  if (Ce > 0) {
    // ---- the Ce-th largest key: most significant byte first ---------------------------------------------------------------
    if (tid == 0) { s_prefix = 0u; s_remaining = Ce; }
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      s_hist[tid] = 0;
      __syncthreads();
      const uint32_t prefix = s_prefix;
      const uint32_t himask = pass == 0 ? 0u : 0xFFFFFFFFu << (shift + 8);
      for (int v = tid; v < V; v += 256) {
        const uint32_t u = key_image(s_key[v]);
        if ((u & himask) == prefix) atomicAdd(&s_hist[(u >> shift) & 255u], 1);
      }
      __syncthreads();
      if (tid < 64) {   // one wave: the digit D with  #(digit > D) < remaining <= #(digit >= D)
        const int remaining = s_remaining;
        int c4[4], above = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) c4[q] = s_hist[255 - (4 * tid + q)];   // lane t holds digits 255 - 4t .. 252 - 4t, descending
        const int mine = c4[0] + c4[1] + c4[2] + c4[3];
        int incl = mine;                                                    // inclusive prefix over the lanes (descending digits)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int n = __shfl_up(incl, o, 64);
          if (tid >= o) incl += n;
        }
        above = incl - mine;                                                // keys with a digit above this lane's four
        if (above < remaining && remaining <= incl) {                       // exactly one lane
          int a = above, D = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (a < remaining && remaining <= a + c4[q]) { D = 255 - (4 * tid + q); s_remaining = remaining - a; }
            a += c4[q];
          }
          s_prefix = prefix | ((uint32_t)D << shift);
        }
      }
      __syncthreads();
    }
    const uint32_t T = s_prefix;              // image of the Ce-th largest key
    const int n_eq = s_remaining;             // how many keys equal to it belong to the sample (the lowest indices)
    const int n_gt = Ce - n_eq;
    // ---- gather: everything above the threshold, and the ties ---------------------------------------------------------------
    for (int v = tid; v < V; v += 256) {
      const uint32_t u = key_image(s_key[v]);
      if (u > T) {
        const int slot = atomicAdd(&s_cnt[1], 1);
        s_selu[slot] = u; s_seli[slot] = v;
      } else if (u == T) {
        const int slot = atomicAdd(&s_cnt[2], 1);
        if (slot < 256) s_tie[slot] = v;
      }
    }
    __syncthreads();
    {   // the n_eq lowest indices among the ties (ties of 23-bit Gumbel keys are rare: normally one key)
      const int nt = min(s_cnt[2], 256);
      if (tid < nt) {
        const int v = s_tie[tid];
        int r = 0;
        for (int j = 0; j < nt; ++j) r += s_tie[j] < v ? 1 : 0;
        if (r < n_eq) { s_selu[n_gt + r] = T; s_seli[n_gt + r] = v; }
      }
    }
    __syncthreads();
    // ---- order of the draws: descending key, equal keys by ascending index ----------------------------------------------------
    for (int i = tid; i < Ce; i += 256) {
      const uint32_t ui = s_selu[i];
      const int vi = s_seli[i];
      int r = 0;
      for (int j = 0; j < Ce; ++j) {
        const uint32_t uj = s_selu[j];
        r += (uj > ui || (uj == ui && s_seli[j] < vi)) ? 1 : 0;
      }
      out[r] = (int64_t)vi;
    }
  }
  for (int c = Ce + tid; c < C; c += 256) out[c] = -1;   // fewer than C items with non-zero probability are left
  if (short_flag != nullptr && Ce < C && tid == 0) *short_flag = 1;   // (every writer stores the same byte)
  if (tid == 0) out[C] = g;
}

extern "C" int b4r_sample_candidates_flagged(const float* logp, int32_t V, const int64_t* exclude, int32_t E, const int64_t* gt,
                                             int32_t R, int32_t C, uint64_t seed, int64_t* cand, uint8_t* short_flag,
                                             b4r_stream_t stream);
extern "C" int b4r_sample_candidates(const float* logp, int32_t V, const int64_t* exclude, int32_t E, const int64_t* gt,
                                     int32_t R, int32_t C, uint64_t seed, int64_t* cand, b4r_stream_t stream) {
  return b4r_sample_candidates_flagged(logp, V, exclude, E, gt, R, C, seed, cand, nullptr, stream);
}
extern "C" int b4r_sample_candidates_flagged(const float* logp, int32_t V, const int64_t* exclude, int32_t E, const int64_t* gt,
                                             int32_t R, int32_t C, uint64_t seed, int64_t* cand, uint8_t* short_flag,
                                             b4r_stream_t stream) {
  B4R_CHECK_ARG(logp && cand && (exclude || E == 0), B4R_E_BADARG, "b4r_sample_candidates: null argument");
  B4R_CHECK_ARG(V > 0 && R > 0 && C > 0 && E >= 0 && C <= V && C <= SAMPLE_MAX_C, B4R_E_SHAPE, "b4r_sample_candidates: bad shape");
  const size_t lds = (size_t)V * sizeof(float);
  B4R_CHECK_ARG(lds <= 150 * 1024, B4R_E_SHAPE, "b4r_sample_candidates: vocabulary %d does not fit the LDS (150 KB of keys)", V);
  { int rc = b4r_raise_lds((const void*)sample_candidates_kernel, lds, "b4r_sample_candidates"); if (rc) return rc; }
  hipLaunchKernelGGL(sample_candidates_kernel, dim3(R), dim3(256), lds, (hipStream_t)stream, logp, V, exclude, E, gt, C,
                     (uint32_t)seed, (uint32_t)(seed >> 32), cand, short_flag);
  B4R_CHECK_LAUNCH("b4r_sample_candidates");
  return B4R_OK;
}


// -----------------------------------------------------------------------------------------------------------
// The last encoder layer's feed-forward half on the rows the masked-LM head gathers only, for every hidden size the resident block
// (b4r_ffn_rx.hip) does not cover: B4R_FLAG_HEAD_ROWS_ONLY with the dense products on COMPACT [B*P, .] operands (b4r_model.hip).
// Compact row m = masked-LM slot m reads / writes sequence row (m / P) * L + clamp(position[m]) (padded slots gather position 0, as
// tfm MaskedLM does, bert4rec_model.py:143: their compact rows repeat a row with the same values).  Dropout decisions are indexed by
// the SEQUENCE row, so the compact path draws the masks the dense path would.
// -----------------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int64_t slot_row(const int64_t* pos, int m, int L, int P) {
  int64_t q = pos[m];
  q = q < 0 ? 0 : (q >= L ? L - 1 : q);
  return (int64_t)(m / P) * L + q;
}
// ac[m] = a[row(m)], bc[m] = b[row(m)] ([., H] rows; b may be NULL), s0c / s1c [m] = s0 / s1 [row(m)] (may be NULL)
__global__ __launch_bounds__(256) void slot_rows_gather_kernel(const float* a, const float* b, const float* s0, const float* s1,
                                                               const int64_t* pos, int L, int P, int M, int H, float* ac, float* bc,
                                                               float* s0c, float* s1c) {
  const int q4 = H >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * q4) return;
  const int m = (int)(idx / q4), c = 4 * (int)(idx % q4);
  const int64_t row = slot_row(pos, m, L, P);
  *reinterpret_cast<f32x4*>(ac + (int64_t)m * H + c) = *reinterpret_cast<const f32x4*>(a + row * H + c);
  if (b != nullptr) *reinterpret_cast<f32x4*>(bc + (int64_t)m * H + c) = *reinterpret_cast<const f32x4*>(b + row * H + c);
  if (c == 0 && s0 != nullptr) { s0c[m] = s0[row]; s1c[m] = s1[row]; }
}
// dst[m] = dropout(src[m]) with the decisions of sequence row row(m)
__global__ __launch_bounds__(256) void slot_rows_drop_kernel(const float* src, const int64_t* pos, int L, int P, int M, int H,
                                                             DropArgs drop, float* dst) {
  const int q4 = H >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)M * q4) return;
  const int m = (int)(idx / q4), c = 4 * (int)(idx % q4);
  const DropCtx dctx = b4r_drop_ctx(drop);
  const int64_t row = slot_row(pos, m, L, P);
  *reinterpret_cast<f32x4*>(dst + (int64_t)m * H + c) =
      b4r_drop4(dctx, *reinterpret_cast<const f32x4*>(src + (int64_t)m * H + c), (uint64_t)row * (uint64_t)H + (uint64_t)c);
}
// z[m] = res[m] + dropout(y[m]); mean / rstd [m]; out[row(m)] = LayerNorm(z[m])   (the thread layout of ln_fwd_kernel)
struct SlotTailP {
  const float* y; const float* res; const int64_t* pos; int L, P;
  const float* gamma; const float* beta; float eps; DropArgs drop;
  float* z; float* mean; float* rstd; float* out; float* outc;   // out [N, H] by sequence row, outc [M, H] compact (either may be NULL)
  int rows, H;
};
template <int LPR, int NV>
__global__ __launch_bounds__(256) void slot_rows_tail_kernel(SlotTailP p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane % LPR, slot = lane / LPR;
  const int64_t m = ((int64_t)blockIdx.x * 4 + wave) * RPW + slot;
  if (m >= p.rows) return;
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  const int64_t row = slot_row(p.pos, (int)m, p.L, p.P);
  f32x4 x[NV];
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + sub) * 4;
    const f32x4 y = b4r_drop4(dctx, *reinterpret_cast<const f32x4*>(p.y + m * p.H + c), (uint64_t)row * (uint64_t)p.H + (uint64_t)c);
    x[v] = *reinterpret_cast<const f32x4*>(p.res + m * p.H + c) + y;
    s += (x[v][0] + x[v][1]) + (x[v][2] + x[v][3]);
  }
  const float mean = group_sum<LPR>(s) / (float)p.H;
  float q = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = x[v][e] - mean; q += d * d; }
  const float rstd = rsqrtf(group_sum<LPR>(q) / (float)p.H + p.eps);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = (v * LPR + sub) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + c), b = *reinterpret_cast<const f32x4*>(p.beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float inv = rstd * g[e];
      o[e] = x[v][e] * inv + (b[e] - mean * inv);
    }
    *reinterpret_cast<f32x4*>(p.z + m * p.H + c) = x[v];
    if (p.out != nullptr) *reinterpret_cast<f32x4*>(p.out + row * p.H + c) = o;
    if (p.outc != nullptr) *reinterpret_cast<f32x4*>(p.outc + m * p.H + c) = o;
  }
  if (sub == 0) { p.mean[m] = mean; p.rstd[m] = rstd; }
}
}  // namespace

int b4r_slot_rows_gather(const float* a, const float* b, const float* s0, const float* s1, const int64_t* pos, int L, int P, int M, int H,
                         float* ac, float* bc, float* s0c, float* s1c, hipStream_t s) {
  hipLaunchKernelGGL(slot_rows_gather_kernel, dim3(b4r_cdiv((int64_t)M * (H / 4), 256)), dim3(256), 0, s, a, b, s0, s1, pos, L, P, M, H, ac,
                     bc, s0c, s1c);
  B4R_CHECK_LAUNCH("last layer on the head's rows: gather");
  return B4R_OK;
}
int b4r_slot_rows_drop(const float* src, const int64_t* pos, int L, int P, int M, int H, const DropArgs& drop, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(slot_rows_drop_kernel, dim3(b4r_cdiv((int64_t)M * (H / 4), 256)), dim3(256), 0, s, src, pos, L, P, M, H, drop, dst);
  B4R_CHECK_LAUNCH("last layer on the head's rows: dropout of the gradient");
  return B4R_OK;
}
int b4r_slot_rows_tail(const float* y, const float* res, const int64_t* pos, int L, int P, int M, int H, const float* gamma, const float* beta,
                       float eps, const DropArgs& drop, float* z, float* mean, float* rstd, float* out, float* outc, hipStream_t s) {
  SlotTailP p{y, res, pos, L, P, gamma, beta, eps, drop, z, mean, rstd, out, outc, M, H};
#define SLOT_TAIL_CASE(LPR_, NV_) \
  hipLaunchKernelGGL((slot_rows_tail_kernel<LPR_, NV_>), dim3(b4r_cdiv(M, 4 * (64 / LPR_))), dim3(256), 0, s, p)
  switch (H) {
    case 32: SLOT_TAIL_CASE(8, 1); break;
    case 64: SLOT_TAIL_CASE(16, 1); break;
    case 128: SLOT_TAIL_CASE(32, 1); break;
    case 256: SLOT_TAIL_CASE(64, 1); break;
    case 512: SLOT_TAIL_CASE(64, 2); break;
    case 1024: SLOT_TAIL_CASE(64, 4); break;
    default: b4r_set_error("last layer on the head's rows: hidden size %d not supported", H); return B4R_E_SHAPE;
  }
#undef SLOT_TAIL_CASE
  B4R_CHECK_LAUNCH("last layer on the head's rows: dropout + residual + LayerNorm");
  return B4R_OK;
}
