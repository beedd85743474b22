// The feed-forward half of an encoder layer for hidden sizes 128 and 256 (the reference's *_128.json / *_256.json configurations, among
// them ml-1m_128.json of its own ML-1M example and ml-20m_256.json): ONE launch forward, the [N, inner] tensor never leaves the chip
// unless a backward will want it.  Reference: tfm TransformerEncoderBlock as built at
// bert4rec/models/components/networks/bert4rec_encoder.py:136-147, called :220-222 --
//     x2 = output_layer_norm(x1 + dropout(gelu(x1 . W1 + b1) . W2 + b2))        (erf GELU, post-LN)
//
// Hidden size 64 keeps W1 and W2 whole in LDS (b4r_ffn_rx.hip).  At 128 / 256 they are 0.5 / 2 MB as bf16 hi / lo images, so the roles
// turn round, on the skeleton of the wide masked-LM head (b4r_head32.hip, head32w_*): a wave keeps 32 TOKEN rows of x1 in registers as
// B operands and the whole [hidden, 32 tokens] output accumulator with them, and the weights stream past in chunks of 32 inner units
// through a two-deep LDS ring filled by LDS-DMA from records packed once per launch (ffn32w_pack_kernel):
//     unit A(c) = W1^T rows 32c .. 32c+31 as panel images [inner unit][hidden] + b1 of the chunk      (row reads: the first product)
//     unit B(c) = W2   rows 32c .. 32c+31 as panel images [inner unit][hidden]                        (transposed reads: the second)
// One step = chunk c:   S^T(c+1) = b1 + A(c+1) . x1^T   (6 NP matrix instructions, three-term bf16 split)
//                       g = gelu(S^T(c)) as bf16 hi / lo, accumulator registers -> B operand (b4r_tile32.h: no LDS round trip)
//                       acc^T += B(c)^T . g            (6 NP matrix instructions, three terms)
// with the GELU of chunk c cut into 64 sub-slices of 4-6 vector instructions, one (NP = 8) or two (NP = 4) behind each matrix
// instruction (the waves issue in order: see the notes in b4r_head32.hip).  The second product runs k-step 0 over all panels first, so
// the GELU of accumulator registers 8..15 still overlaps its first half.  Epilogue in registers: + b2, dropout, + x1, LayerNorm.
#include "b4r_tile32.h"

// timing experiments only (tools/build_variant.sh x b4r_ffn32w.hip -DF32W_EXP=n): 1 no GELU arithmetic, 2 no first product, 4 no second
// product, 8 no copies inside the loop, 16 no barrier inside the loop, 32 fragment reads once per step
#ifndef F32W_EXP
#define F32W_EXP 0
#endif

namespace {

constexpr int F32W_SIDE = 256;                                                                // bytes behind unit A: b1 of the chunk (32 floats)
__host__ __device__ constexpr int f32w_unit_a(int np) { return np * P_TILE + F32W_SIDE; }
__host__ __device__ constexpr int f32w_unit_b(int np) { return np * P_TILE; }
__host__ __device__ constexpr int f32w_rec(int np) { return f32w_unit_a(np) + f32w_unit_b(np); }

// ---------------------------------------------------------------------------------------------------------------------------
// pack: one workgroup per chunk of 32 inner units
// ---------------------------------------------------------------------------------------------------------------------------
struct F32wPackP { const float* W1; const float* b1; const float* W2; int I; char* dst; };

template <int NP>
__device__ __forceinline__ void f32w_put4(char* unit, int row, int q, const f32x4 x) {
  bf16x4 hh, ll;
  b4r_split4(x, hh, ll);
  char* d8 = unit + (q >> 3) * P_TILE + p_chunk(row, (q & 7) >> 1) + 8 * (q & 1);
  *reinterpret_cast<bf16x4*>(d8) = hh;
  *reinterpret_cast<bf16x4*>(d8 + P_IMG) = ll;
}
template <int NP>
__global__ __launch_bounds__(256) void ffn32w_pack_kernel(F32wPackP p) {
  constexpr int H = 32 * NP;
  __shared__ float t[32][H + 1];
  const int c = blockIdx.x;
  char* rec = p.dst + (int64_t)c * f32w_rec(NP);
  // W1 [H, I]: column chunk -> t[j][k]  (coalesced 128-byte reads per k)
  for (int f = threadIdx.x; f < 32 * H; f += 256) {
    const int k = f >> 5, j = f & 31;
    t[j][k] = p.W1[(int64_t)k * p.I + 32 * c + j];
  }
  // unit B: rows of W2 [I, H]
  char* ub = rec + f32w_unit_a(NP);
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int f = threadIdx.x + 256 * it, row = f / (8 * NP), q = f % (8 * NP);
    f32w_put4<NP>(ub, row, q, *reinterpret_cast<const f32x4*>(p.W2 + (int64_t)(32 * c + row) * H + 4 * q));
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int f = threadIdx.x + 256 * it, row = f / (8 * NP), q = f % (8 * NP);
    f32w_put4<NP>(rec, row, q, (f32x4){t[row][4 * q], t[row][4 * q + 1], t[row][4 * q + 2], t[row][4 * q + 3]});
  }
  if (threadIdx.x < 64) {
    float* side = reinterpret_cast<float*>(rec + NP * P_TILE);
    side[threadIdx.x] = threadIdx.x < 32 ? p.b1[32 * c + threadIdx.x] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LDS-DMA (see b4r_head32.hip for why this is inline asm and why the waits are counted by hand)
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void f32w_dma16(const char* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ void f32w_dma4(const char* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ unsigned f32w_lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
// the NP panel tiles of a unit: 4 NP pieces of 1 KB, 4 NP / WAVES per wave; SIDE: wave 0 also copies the 256 bytes behind them
template <int NP, int WAVES, bool SIDE>
__device__ __forceinline__ void f32w_issue(const char* src, char* slot, int wave, int lane) {
  constexpr int G = 4 * NP / WAVES;
  const unsigned dst = f32w_lds_addr(slot);
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const int piece = wave * G + q;
    f32w_dma16(src + piece * 1024 + lane * 16, dst + piece * 1024);
  }
  if (SIDE && wave == 0) f32w_dma4(src + NP * P_TILE + lane * 4, dst + NP * P_TILE);
}
__device__ __forceinline__ void f32w_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------------------
// erf-GELU of two accumulator registers in eight stages of 2-6 vector instructions (Abramowitz & Stegun 7.1.26, the polynomial of
// b4r_erf_as; the argument is pre-scaled by sqrt(log2 e) so that exp(-u^2) is ONE v_exp_f32 of -(v^2)), then the bf16 hi / lo split of
// the pair.  Every stage pins its results with an empty asm statement: pure arithmetic is otherwise sunk to its first use.
// ---------------------------------------------------------------------------------------------------------------------------
struct GeluPair { float x[2], v[2], d[2], t[2], e[2], q[2], g[2], keep[2]; };
constexpr double F32W_S = 1.2011224087864498;   // sqrt(log2 e)
// gv: the gelu values in the accumulator's register order (KEEP: they leave as f through the staged store) -- or, qdst != NULL (a
// compile-time fact after inlining), every second pair stores its 16 bytes to qdst[8 (q >> 1) ..] at once
__device__ __forceinline__ void f32w_gelu_stage(int st, int q, const f32x16& S, GeluPair& a, uint32_t (&hw)[8], uint32_t (&lw)[8],
                                                f32x16& gv, float* qdst) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    switch (st) {
      case 0:
        a.x[e] = S[2 * q + e];
        a.v[e] = a.x[e] * (float)(0.70710678118654752440 * F32W_S);
        a.d[e] = fmaf((float)(0.3275911 / F32W_S), fabsf(a.v[e]), 1.0f);
        asm volatile("" : "+v"(a.v[e]), "+v"(a.d[e]));
        break;
      case 1:
        a.t[e] = __builtin_amdgcn_rcpf(a.d[e]);
        asm volatile("" : "+v"(a.t[e]));
        break;
      case 2:
        a.e[e] = __builtin_amdgcn_exp2f(-(a.v[e] * a.v[e]));
        asm volatile("" : "+v"(a.e[e]));
        break;
      case 3:
        a.q[e] = fmaf(fmaf(a.t[e], 1.061405429f, -1.453152027f), a.t[e], 1.421413741f);
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 4:
        a.q[e] = fmaf(fmaf(a.q[e], a.t[e], -0.284496736f), a.t[e], 0.254829592f);
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 5:
        a.q[e] = fmaf(-(a.q[e] * a.t[e]), a.e[e], 1.0f);      // erf(|u|)
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 6: {
        const float er = copysignf(a.q[e], a.v[e]), hx = 0.5f * a.x[e];
        a.g[e] = fmaf(hx, er, hx);
        asm volatile("" : "+v"(a.g[e]));
        break;
      }
      default: break;
    }
  }
  if (st == 7) {
    b4r_split_pair(a.g[0], a.g[1], hw[q], lw[q]);
    asm volatile("" : "+v"(hw[q]), "+v"(lw[q]));
    if (qdst == nullptr) { gv[2 * q] = a.g[0]; gv[2 * q + 1] = a.g[1]; }
    else if ((q & 1) == 0) { a.keep[0] = a.g[0]; a.keep[1] = a.g[1]; }
    else *reinterpret_cast<f32x4*>(qdst + 8 * (q >> 1)) = (f32x4){a.keep[0], a.keep[1], a.g[0], a.g[1]};
  }
}
__device__ __forceinline__ bf16x8 f32w_frag(const uint32_t (&w)[8], int s) {
  return __builtin_bit_cast(bf16x8, (b4r_u32x4){w[4 * s], w[4 * s + 1], w[4 * s + 2], w[4 * s + 3]});
}

// One step.  a_nxt: unit A of chunk c + 1 (row reads), b_cur: unit B of chunk c (transposed reads); S = S^T(c) complete, Sn = b1 of
// chunk c + 1 on entry and S^T(c + 1) on exit.
template <int NP>
__device__ __forceinline__ void f32w_step(const char* a_nxt, const char* b_cur, const Lane32& lk, const bf16x8 (&xh)[NP][2],
                                          const bf16x8 (&xl)[NP][2], f32x16& Sn, f32x16 (&acc)[NP], const f32x16& S, f32x16& gv, float* qdst) {
  constexpr int NT = 2 * NP, PER = 8 / NP;
  static_assert(PER >= 1 && 32 <= PER * 6 * NP && 64 <= PER * 9 * NP, "GELU sub-slices must meet the second product's operands");
  GeluPair gp;
  uint32_t ghw[8], glw[8];
  int m = 0;
  auto behind = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int ss = PER * m + u;
      if (!(F32W_EXP & 1) && ss < 64) f32w_gelu_stage(ss & 7, ss >> 3, S, gp, ghw, glw, gv, qdst);
    }
    ++m;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_l = [&](int j, bf16x8& ah, bf16x8& al) __attribute__((always_inline)) {
    const char* a = a_nxt + (j >> 1) * P_TILE + lk.rowc[j & 1];
    ah = row_at(a); al = row_at(a + P_IMG);
  };
  auto load_f = [&](int g, bf16x8& ah, bf16x8& al) __attribute__((always_inline)) {   // group g = s * NP + p
    const int s = g / NP, pp = g % NP;
    const char* a = b_cur + pp * P_TILE;
    ah = tr_pair(a + lk.trp[s][0], a + lk.trp[s][1]);
    al = tr_pair(a + P_IMG + lk.trp[s][0], a + P_IMG + lk.trp[s][1]);
  };
  bf16x8 ah, al, nh, nl;
  load_l(0, ah, al);
  __builtin_amdgcn_sched_barrier(0);
  if (F32W_EXP & 1) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { ghw[q] = __builtin_bit_cast(uint32_t, S[2 * q]); glw[q] = __builtin_bit_cast(uint32_t, S[2 * q + 1]); }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!(F32W_EXP & 32)) { if (j + 1 < NT) load_l(j + 1, nh, nl); else load_f(0, nh, nl); }
    if (!(F32W_EXP & 2)) Sn = mfma32(al, xh[j >> 1][j & 1], Sn);
    behind();
    if (!(F32W_EXP & 2)) Sn = mfma32(ah, xl[j >> 1][j & 1], Sn);
    behind();
    if (!(F32W_EXP & 2)) Sn = mfma32(ah, xh[j >> 1][j & 1], Sn);
    behind();
    if (!(F32W_EXP & 32)) { ah = nh; al = nl; }
  }
#pragma unroll
  for (int g = 0; g < NT; ++g) {
    const int s = g / NP, pp = g % NP;
    if (!(F32W_EXP & 32)) { if (g + 1 < NT) load_f(g + 1, nh, nl); }
    const bf16x8 gh = f32w_frag(ghw, s), gl = f32w_frag(glw, s);
    if (!(F32W_EXP & 4)) acc[pp] = mfma32(al, gh, acc[pp]);
    behind();
    if (!(F32W_EXP & 4)) acc[pp] = mfma32(ah, gl, acc[pp]);
    behind();
    if (!(F32W_EXP & 4)) acc[pp] = mfma32(ah, gh, acc[pp]);
    behind();
    if (!(F32W_EXP & 32)) { ah = nh; al = nl; }
  }
}

// accumulator registers 4 g4 .. 4 g4 + 3 of lane half h = inner units (or features) 8 g4 + 4 h .. + 3 of the 32-row block
__device__ __forceinline__ f32x4 f32w_quad(const f32x16& v, int g4) { return (f32x4){v[4 * g4], v[4 * g4 + 1], v[4 * g4 + 2], v[4 * g4 + 3]}; }
// A wave's [32 inner units x 32 tokens] accumulator tile -> 32 x 128 bytes of a [N, I] tensor as whole 128-byte row segments: through
// 4.5 KB of the wave's own LDS ([token][36 floats]); a wave-instruction then stores 8 rows x 128 contiguous bytes.  (Straight from the
// accumulator layout a lane holds 4 x 16 bytes of ITS token's row: 32 rows x 32 bytes per instruction -- 420 MB of f and fpre cost the
// hidden-256 forward 135 us that way.)
// MEASURED (N = 51 200, us): hidden 256 forward keeping f / fpre 394 -> 323, backward 414 -> 403; hidden 128 (row pitch 2 KB, a step of
// half the length) 102 -> 110 and 103 -> 115: there the lanes store their own 16-byte pieces (STAGED = false).
constexpr int F32W_STG_LD = 36, F32W_STG_BYTES = 32 * F32W_STG_LD * 4;
template <bool STAGED>
__device__ __forceinline__ void f32w_store_tile(float* stg, const f32x16& v, float* dst, int ld, int rows_left, int lane, int r, int h) {
  if (!STAGED) {
    if (r < rows_left) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<f32x4*>(dst + (int64_t)r * ld + 8 * g4 + 4 * h) = f32w_quad(v, g4);
    }
    return;
  }
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) *reinterpret_cast<f32x4*>(stg + r * F32W_STG_LD + 8 * g4 + 4 * h) = f32w_quad(v, g4);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = (lane >> 3) + 8 * k;
    const f32x4 q = *reinterpret_cast<const f32x4*>(stg + row * F32W_STG_LD + 4 * (lane & 7));
    if (row < rows_left) *reinterpret_cast<f32x4*>(dst + (int64_t)row * ld + 4 * (lane & 7)) = q;
  }
  asm volatile("" ::: "memory");
}

struct F32wP {
  const float* x1; int N;
  const char* recs; int n_chunks;
  const float* b2; const float* gamma; const float* beta; float eps;
  DropArgs drop;
  float* z2; float* x2; float* mean2; float* rstd2;
  float* f; float* fpre; int I;            // optional [N, I]: gelu output and pre-activation for a backward
};


// KEEP: the launch also writes f = gelu(.) and the pre-activation [N, I] (the backward's inputs)
template <int NP, int WAVES, bool KEEP>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void ffn32w_fwd_kernel(F32wP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_f32w[];
  constexpr int H = 32 * NP, UA = f32w_unit_a(NP), UB = f32w_unit_b(NP), REC = UA + UB, ROWS = 32 * WAVES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane32 lk = lane32(lane);
  const int r = lk.r, h = lk.h;
  const int m0 = blockIdx.x * ROWS + 32 * wave, m = m0 + r;   // (a wave whose rows all lie beyond N still runs: barriers)
  const int64_t mc = min(m, p.N - 1);
  float* stg = reinterpret_cast<float*>(smem_f32w + 2 * REC + wave * F32W_STG_BYTES);   // KEEP: the wave's store staging
  bf16x8 xh[NP][2], xl[NP][2];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int s = 0; s < 2; ++s) split8(load8(p.x1 + mc * H + 32 * pp + 16 * s + 8 * h), xh[pp][s], xl[pp][s]);
  __builtin_amdgcn_sched_barrier(0);
  const int n = p.n_chunks;
  auto slot_a = [&](int i) __attribute__((always_inline)) { return smem_f32w + (i & 1) * UA; };
  auto slot_b = [&](int i) __attribute__((always_inline)) { return smem_f32w + 2 * UA + (i & 1) * UB; };
  auto issue_a = [&](int i) __attribute__((always_inline)) {
    f32w_issue<NP, WAVES, true>(p.recs + (int64_t)min(i, n - 1) * REC, slot_a(i), wave, lane);
  };
  auto issue_b = [&](int i) __attribute__((always_inline)) {
    f32w_issue<NP, WAVES, false>(p.recs + (int64_t)min(i, n - 1) * REC + UA, slot_b(i), wave, lane);
  };
  issue_a(0); issue_b(0); issue_a(1);
  f32x16 acc[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) acc[pp] = zero16();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32w_barrier();
  // S^T(0), outside the pipeline
  f32x16 S0 = rows_of(reinterpret_cast<const float*>(slot_a(0) + NP * P_TILE), h), S1;
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) {
    const char* a = slot_a(0) + (j >> 1) * P_TILE + lk.rowc[j & 1];
    S0 = mfma32x3(row_at(a), row_at(a + P_IMG), xh[j >> 1][j & 1], xl[j >> 1][j & 1], S0);
  }
  auto step = [&](int i, f32x16& S, f32x16& Sn) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // unit A(i + 1) and unit B(i): this wave's pieces ...
    if (!(F32W_EXP & 16)) f32w_barrier();                     // ... every wave's; and every wave is done with A(i), B(i - 1)
    if (!(F32W_EXP & 8)) { issue_a(i + 2); issue_b(i + 1); }
    __builtin_amdgcn_sched_barrier(0);
    if (KEEP) f32w_store_tile<(NP > 4)>(stg, S, p.fpre + (int64_t)m0 * p.I + 32 * i, p.I, p.N - m0, lane, r, h);
    Sn = rows_of(reinterpret_cast<const float*>(slot_a(i + 1) + NP * P_TILE), h);
    f32x16 gv;
    constexpr bool STAGED = NP > 4;
    f32w_step<NP>(slot_a(i + 1), slot_b(i), lk, xh, xl, Sn, acc, S, gv, (KEEP && !STAGED) ? p.f + mc * p.I + 32 * i + 4 * h : nullptr);
    if (KEEP && STAGED) f32w_store_tile<true>(stg, gv, p.f + (int64_t)m0 * p.I + 32 * i, p.I, p.N - m0, lane, r, h);
  };
  for (int i = 0; i < n; i += 2) {
    step(i, S0, S1);
    if (i + 1 < n) step(i + 1, S1, S0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // epilogue: z2 = x1 + dropout(acc + b2), x2 = LayerNorm(z2)
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  float sum = 0.f;
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int c = 32 * pp + 8 * g4 + 4 * h;
      f32x4 y = f32w_quad(acc[pp], g4) + *reinterpret_cast<const f32x4*>(p.b2 + c);
      y = b4r_drop4(dctx, y, (uint64_t)mc * (uint64_t)H + (uint64_t)c);
      y = y + *reinterpret_cast<const f32x4*>(p.x1 + mc * H + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[pp][4 * g4 + e] = y[e];
      sum += (y[0] + y[1]) + (y[2] + y[3]);
    }
  sum += other_half(sum, h);
  const float mean = sum / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int t = 0; t < 16; ++t) { const float d = acc[pp][t] - mean; sq += d * d; }
  sq += other_half(sq, h);
  const float rstd = rsqrtf(sq / (float)H + p.eps);
  if (m < p.N) {
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c = 32 * pp + 8 * g4 + 4 * h;
        const f32x4 z = f32w_quad(acc[pp], g4);
        if (p.z2 != nullptr) *reinterpret_cast<f32x4*>(p.z2 + (int64_t)m * H + c) = z;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c), be = *reinterpret_cast<const f32x4*>(p.beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float inv = rstd * gm[e];
          o[e] = z[e] * inv + (be[e] - mean * inv);
        }
        *reinterpret_cast<f32x4*>(p.x2 + (int64_t)m * H + c) = o;
      }
    if (h == 0) {
      if (p.mean2 != nullptr) p.mean2[m] = mean;
      if (p.rstd2 != nullptr) p.rstd2[m] = rstd;
    }
  }
}

template <int NP, int WAVES, bool KEEP>
int f32w_launch_fwd(const F32wPackP& pk, const F32wP& p, hipStream_t stream) {
  hipLaunchKernelGGL(ffn32w_pack_kernel<NP>, dim3(p.n_chunks), dim3(256), 0, stream, pk);
  B4R_CHECK_LAUNCH("wide feed-forward block: weight records");
  const size_t lds = 2 * (size_t)f32w_rec(NP) + ((KEEP && NP > 4) ? (size_t)WAVES * F32W_STG_BYTES : 0);
  int rc = b4r_raise_lds((const void*)ffn32w_fwd_kernel<NP, WAVES, KEEP>, lds, "wide feed-forward block");
  if (rc) return rc;
  hipLaunchKernelGGL((ffn32w_fwd_kernel<NP, WAVES, KEEP>), dim3(b4r_cdiv(p.N, 32 * WAVES)), dim3(64 * WAVES), lds, stream, p);
  B4R_CHECK_LAUNCH("wide feed-forward block forward");
  return B4R_OK;
}


// ===========================================================================================================================
// backward, input-gradient side:  df = (dropmask(dz2) . W2^T) * gelu'(fpre)   and   dx1 = df . W1^T + dz2      (one launch)
// The same skeleton with the units' roles exchanged: a wave keeps 32 token rows of dropmask(dz2) as B operands and dx1^T as its
// accumulator; per chunk c   G^T(c+1) = B(c+1) . dz2d^T   (row reads of unit B),   dz = G^T(c) * gelu'(fpre tile c) (the tile arrives
// in the accumulator's register layout by four 16-byte loads per lane, requested a step ahead), stored as df and split to bf16
// hi / lo,   acc^T += A(c)^T . dz   (transposed reads of unit A).  The weight gradients stay with b4r_gemm_tn_f32 (they sum over
// TOKENS: a different sweep), which reads f, df, x1 and dz2.
// ===========================================================================================================================
struct GeluGradPair { float x[2], v[2], d[2], t[2], e[2], q[2], g[2], keep[2]; };
__device__ __forceinline__ void f32w_dgelu_stage(int st, int q, const f32x16& X, const f32x16& G, GeluGradPair& a, uint32_t (&hw)[8],
                                                 uint32_t (&lw)[8], f32x16& dv, float* qdst) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    switch (st) {
      case 0:
        a.x[e] = X[2 * q + e];
        a.v[e] = a.x[e] * (float)(0.70710678118654752440 * F32W_S);
        a.d[e] = fmaf((float)(0.3275911 / F32W_S), fabsf(a.v[e]), 1.0f);
        asm volatile("" : "+v"(a.v[e]), "+v"(a.d[e]));
        break;
      case 1:
        a.t[e] = __builtin_amdgcn_rcpf(a.d[e]);
        asm volatile("" : "+v"(a.t[e]));
        break;
      case 2:
        a.e[e] = __builtin_amdgcn_exp2f(-(a.v[e] * a.v[e]));
        asm volatile("" : "+v"(a.e[e]));
        break;
      case 3:
        a.q[e] = fmaf(fmaf(a.t[e], 1.061405429f, -1.453152027f), a.t[e], 1.421413741f);
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 4:
        a.q[e] = fmaf(fmaf(a.q[e], a.t[e], -0.284496736f), a.t[e], 0.254829592f);
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 5:
        a.q[e] = fmaf(-(a.q[e] * a.t[e]), a.e[e], 1.0f);      // erf(|u|)
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 6:
        a.q[e] = fmaf(0.5f, copysignf(a.q[e], a.v[e]), 0.5f);  // Phi(x)
        asm volatile("" : "+v"(a.q[e]));
        break;
      case 7:
        a.g[e] = G[2 * q + e] * fmaf(a.x[e] * 0.39894228040143267794f, a.e[e], a.q[e]);   // G * (Phi(x) + x phi(x))
        asm volatile("" : "+v"(a.g[e]));
        break;
      default: break;
    }
  }
  if (st == 8) {
    b4r_split_pair(a.g[0], a.g[1], hw[q], lw[q]);
    asm volatile("" : "+v"(hw[q]), "+v"(lw[q]));
    if (qdst == nullptr) { dv[2 * q] = a.g[0]; dv[2 * q + 1] = a.g[1]; }
    else if ((q & 1) == 0) { a.keep[0] = a.g[0]; a.keep[1] = a.g[1]; }
    else *reinterpret_cast<f32x4*>(qdst + 8 * (q >> 1)) = (f32x4){a.keep[0], a.keep[1], a.g[0], a.g[1]};
  }
}

// One step.  b_nxt: unit B of chunk c + 1 (row reads), a_cur: unit A of chunk c (transposed reads); G = G^T(c), X = fpre tile c.
template <int NP>
__device__ __forceinline__ void f32w_bstep(const char* b_nxt, const char* a_cur, const Lane32& lk, const bf16x8 (&dh)[NP][2],
                                           const bf16x8 (&dl)[NP][2], f32x16& Gn, f32x16 (&acc)[NP], const f32x16& G, const f32x16& X,
                                           f32x16& dv, float* qdst) {
  constexpr int NT = 2 * NP, PER = 8 / NP;
  static_assert(PER >= 1 && 36 <= PER * 6 * NP && 72 <= PER * 9 * NP, "GELU' sub-slices must meet the second product's operands");
  GeluGradPair gp;
  uint32_t ghw[8], glw[8];
  int m = 0;
  auto behind = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int ss = PER * m + u;
      if (ss < 72) f32w_dgelu_stage(ss % 9, ss / 9, X, G, gp, ghw, glw, dv, qdst);
    }
    ++m;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_l = [&](int j, bf16x8& ah, bf16x8& al) __attribute__((always_inline)) {
    const char* a = b_nxt + (j >> 1) * P_TILE + lk.rowc[j & 1];
    ah = row_at(a); al = row_at(a + P_IMG);
  };
  auto load_f = [&](int g, bf16x8& ah, bf16x8& al) __attribute__((always_inline)) {
    const int s = g / NP, pp = g % NP;
    const char* a = a_cur + pp * P_TILE;
    ah = tr_pair(a + lk.trp[s][0], a + lk.trp[s][1]);
    al = tr_pair(a + P_IMG + lk.trp[s][0], a + P_IMG + lk.trp[s][1]);
  };
  bf16x8 ah, al, nh, nl;
  load_l(0, ah, al);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j + 1 < NT) load_l(j + 1, nh, nl); else load_f(0, nh, nl);
    Gn = mfma32(al, dh[j >> 1][j & 1], Gn); behind();
    Gn = mfma32(ah, dl[j >> 1][j & 1], Gn); behind();
    Gn = mfma32(ah, dh[j >> 1][j & 1], Gn); behind();
    ah = nh; al = nl;
  }
#pragma unroll
  for (int g = 0; g < NT; ++g) {
    const int s = g / NP, pp = g % NP;
    if (g + 1 < NT) load_f(g + 1, nh, nl);
    const bf16x8 gh = f32w_frag(ghw, s), gl = f32w_frag(glw, s);
    acc[pp] = mfma32(al, gh, acc[pp]); behind();
    acc[pp] = mfma32(ah, gl, acc[pp]); behind();
    acc[pp] = mfma32(ah, gh, acc[pp]); behind();
    ah = nh; al = nl;
  }
}

// keeps an operand fragment in the vector registers proper (hipcc otherwise parks the 128 operand registers of hidden size 256 beside
// the accumulators in the AGPR half, fills it, and spills 93 registers of the other half's working set)
__device__ __forceinline__ void f32w_pin_v(bf16x8& x) {
  b4r_u32x4 w = __builtin_bit_cast(b4r_u32x4, x);
  asm volatile("" : "+v"(w));
  x = __builtin_bit_cast(bf16x8, w);
}
struct F32wBwdP {
  const float* dz2; int N;
  const char* recs; int n_chunks;
  DropArgs drop;
  const float* fpre; float* df; int I;
  float* dx1;
};

__device__ __forceinline__ f32x16 f32w_tile_rows(const float* src, int h) {   // the tile's 16 values of this lane: src = row base + 32 c
  f32x16 v;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(src + 8 * g4 + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * g4 + e] = q[e];
  }
  return v;
}

template <int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void ffn32w_bwd_kernel(F32wBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_f32w[];
  constexpr int H = 32 * NP, UA = f32w_unit_a(NP), UB = f32w_unit_b(NP), REC = UA + UB, ROWS = 32 * WAVES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane32 lk = lane32(lane);
  const int r = lk.r, h = lk.h;
  const int m0 = blockIdx.x * ROWS + 32 * wave, m = m0 + r;
  const int64_t mc = min(m, p.N - 1);
  float* stg = reinterpret_cast<float*>(smem_f32w + 2 * REC + wave * F32W_STG_BYTES);
  const DropCtx dctx = b4r_drop_ctx(p.drop);
  bf16x8 dh[NP][2], dl[NP][2];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = 32 * pp + 16 * s + 8 * h;
      const f32x4 a = b4r_drop4(dctx, *reinterpret_cast<const f32x4*>(p.dz2 + mc * H + c), (uint64_t)mc * (uint64_t)H + (uint64_t)c);
      const f32x4 b = b4r_drop4(dctx, *reinterpret_cast<const f32x4*>(p.dz2 + mc * H + c + 4), (uint64_t)mc * (uint64_t)H + (uint64_t)(c + 4));
      split8(cat(a, b), dh[pp][s], dl[pp][s]);
      f32w_pin_v(dh[pp][s]); f32w_pin_v(dl[pp][s]);
    }
  __builtin_amdgcn_sched_barrier(0);
  const int n = p.n_chunks;
  auto slot_a = [&](int i) __attribute__((always_inline)) { return smem_f32w + (i & 1) * UA; };
  auto slot_b = [&](int i) __attribute__((always_inline)) { return smem_f32w + 2 * UA + (i & 1) * UB; };
  auto issue_a = [&](int i) __attribute__((always_inline)) {
    f32w_issue<NP, WAVES, false>(p.recs + (int64_t)min(i, n - 1) * REC, slot_a(i), wave, lane);
  };
  auto issue_b = [&](int i) __attribute__((always_inline)) {
    f32w_issue<NP, WAVES, false>(p.recs + (int64_t)min(i, n - 1) * REC + UA, slot_b(i), wave, lane);
  };
  const float* xrow = p.fpre + mc * p.I;
  issue_b(0); issue_a(0); issue_b(1);
  f32x16 X0 = f32w_tile_rows(xrow, h), X1;
  f32x16 acc[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) acc[pp] = zero16();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32w_barrier();
  f32x16 G0 = zero16(), G1;
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) {
    const char* a = slot_b(0) + (j >> 1) * P_TILE + lk.rowc[j & 1];
    G0 = mfma32x3(row_at(a), row_at(a + P_IMG), dh[j >> 1][j & 1], dl[j >> 1][j & 1], G0);
  }
  auto step = [&](int i, f32x16& G, f32x16& Gn, f32x16& X, f32x16& Xn) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // unit B(i + 1), unit A(i), the fpre tile i: this wave's share ...
    f32w_barrier();                                           // ... every wave's; and every wave is done with B(i), A(i - 1)
    issue_b(i + 2); issue_a(i + 1);
    Xn = f32w_tile_rows(xrow + 32 * min(i + 1, n - 1), h);
    __builtin_amdgcn_sched_barrier(0);
    Gn = zero16();
    f32x16 dv;
    constexpr bool STAGED = NP > 4;
    f32w_bstep<NP>(slot_b(i + 1), slot_a(i), lk, dh, dl, Gn, acc, G, X, dv, STAGED ? nullptr : p.df + mc * p.I + 32 * i + 4 * h);
    if (STAGED) f32w_store_tile<true>(stg, dv, p.df + (int64_t)m0 * p.I + 32 * i, p.I, p.N - m0, lane, r, h);
  };
  for (int i = 0; i < n; i += 2) {
    step(i, G0, G1, X0, X1);
    if (i + 1 < n) step(i + 1, G1, G0, X1, X0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (m < p.N) {
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int c = 32 * pp + 8 * g4 + 4 * h;
        *reinterpret_cast<f32x4*>(p.dx1 + (int64_t)m * H + c) =
            f32w_quad(acc[pp], g4) + *reinterpret_cast<const f32x4*>(p.dz2 + (int64_t)m * H + c);
      }
  }
}

template <int NP, int WAVES>
int f32w_launch_bwd(const F32wPackP* pk, const F32wBwdP& p, hipStream_t stream) {
  if (pk != nullptr) {
    hipLaunchKernelGGL(ffn32w_pack_kernel<NP>, dim3(p.n_chunks), dim3(256), 0, stream, *pk);
    B4R_CHECK_LAUNCH("wide feed-forward block: weight records");
  }
  const size_t lds = 2 * (size_t)f32w_rec(NP) + (NP > 4 ? (size_t)WAVES * F32W_STG_BYTES : 0);
  int rc = b4r_raise_lds((const void*)ffn32w_bwd_kernel<NP, WAVES>, lds, "wide feed-forward block");
  if (rc) return rc;
  hipLaunchKernelGGL((ffn32w_bwd_kernel<NP, WAVES>), dim3(b4r_cdiv(p.N, 32 * WAVES)), dim3(64 * WAVES), lds, stream, p);
  B4R_CHECK_LAUNCH("wide feed-forward block backward (df, dx1)");
  return B4R_OK;
}

}  // namespace

bool b4r_ffn32w_supported(int H, int I) {
  static const bool on = !(getenv("B4R_FFN32W") && atoi(getenv("B4R_FFN32W")) == 0);
  return on && (H == 128 || H == 256) && I >= 64 && I % 32 == 0 && b4r_get_gemm_mode() == B4R_GEMM_BF16X3;
}
int64_t b4r_ffn32w_rec_floats(int H, int I) { return ((int64_t)(I / 32) * f32w_rec(H / 32) + 3) / 4; }

// forward of the block described by d (x1 given; no row list); recs: b4r_ffn32w_rec_floats floats; f / fpre: optional [N, I] stores
int b4r_ffn32w_fwd(const b4r_ffn_desc* d, float* recs, float* f, float* fpre, hipStream_t stream) {
  F32wPackP pk{d->W1, d->b1, d->W2, d->I, reinterpret_cast<char*>(recs)};
  F32wP p{};
  p.x1 = d->x1; p.N = d->N; p.recs = reinterpret_cast<const char*>(recs); p.n_chunks = d->I / 32;
  p.b2 = d->b2; p.gamma = d->ln_gamma; p.beta = d->ln_beta; p.eps = d->ln_eps;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  p.z2 = d->z2; p.x2 = d->x2; p.mean2 = d->mean2; p.rstd2 = d->rstd2;
  p.f = f; p.fpre = fpre; p.I = d->I;
  B4R_CHECK_ARG((f == nullptr) == (fpre == nullptr), B4R_E_BADARG, "b4r_ffn32w_fwd: f and fpre come together");
  if (f != nullptr) return d->H == 128 ? f32w_launch_fwd<4, 8, true>(pk, p, stream) : f32w_launch_fwd<8, 4, true>(pk, p, stream);
  return d->H == 128 ? f32w_launch_fwd<4, 8, false>(pk, p, stream) : f32w_launch_fwd<8, 4, false>(pk, p, stream);
}

// df [N, I] and dx1 [N, H] (residual included) from dz2 and the forward's fpre; recs as left by b4r_ffn32w_fwd (records_ready) or packed here
int b4r_ffn32w_bwd(const b4r_ffn_desc* d, float* recs, const float* fpre, float* df, float* dx1, bool records_ready, hipStream_t stream) {
  F32wPackP pk{d->W1, d->b1, d->W2, d->I, reinterpret_cast<char*>(recs)};
  F32wBwdP p{};
  p.dz2 = d->dz2; p.N = d->N; p.recs = reinterpret_cast<const char*>(recs); p.n_chunks = d->I / 32;
  p.drop = b4r_make_drop(d->rng, d->drop_stream, d->drop_rate, 1);
  p.fpre = fpre; p.df = df; p.I = d->I; p.dx1 = dx1;
  return d->H == 128 ? f32w_launch_bwd<4, 8>(records_ready ? nullptr : &pk, p, stream)
                     : f32w_launch_bwd<8, 4>(records_ready ? nullptr : &pk, p, stream);
}

extern "C" int32_t b4r_ffn_wide_supported(int32_t hidden_size, int32_t inner_dim) { return b4r_ffn32w_supported(hidden_size, inner_dim) ? 1 : 0; }
extern "C" int64_t b4r_ffn_wide_scratch_floats(int32_t hidden_size, int32_t inner_dim) {
  return b4r_ffn32w_supported(hidden_size, inner_dim) ? b4r_ffn32w_rec_floats(hidden_size, inner_dim) : 0;
}
namespace {
int f32w_check(const b4r_ffn_desc* d, const char* who) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "%s: null descriptor", who);
  B4R_CHECK_ARG(b4r_ffn32w_supported(d->H, d->I), B4R_E_SHAPE, "%s: hidden %d / inner %d not supported (hidden 128 or 256, inner a multiple of 32)",
                who, d->H, d->I);
  B4R_CHECK_ARG(d->N > 0 && d->W1 && d->b1 && d->W2 && d->scratch, B4R_E_BADARG, "%s: N, W1, b1, W2 and scratch are required", who);
  B4R_CHECK_ARG(d->rows == nullptr && d->slot_positions == nullptr, B4R_E_BADARG, "%s: no row list in the wide block", who);
  B4R_CHECK_ARG(b4r_aligned16(d->scratch), B4R_E_BADARG, "%s: scratch must be 16-byte aligned", who);
  return B4R_OK;
}
}  // namespace
extern "C" int b4r_ffn_wide_fwd(const b4r_ffn_desc* d, float* f, float* fpre, b4r_stream_t stream) {
  int rc = f32w_check(d, "b4r_ffn_wide_fwd");
  if (rc) return rc;
  B4R_CHECK_ARG(d->x1 && d->b2 && d->ln_gamma && d->ln_beta && d->x2, B4R_E_BADARG, "b4r_ffn_wide_fwd: x1, b2, ln_gamma, ln_beta and x2 are required");
  return b4r_ffn32w_fwd(d, d->scratch, f, fpre, (hipStream_t)stream);
}
extern "C" int b4r_ffn_wide_bwd(const b4r_ffn_desc* d, const float* fpre, float* df, float* dx1, int32_t records_ready, b4r_stream_t stream) {
  int rc = f32w_check(d, "b4r_ffn_wide_bwd");
  if (rc) return rc;
  B4R_CHECK_ARG(d->dz2 && fpre && df && dx1, B4R_E_BADARG, "b4r_ffn_wide_bwd: dz2, fpre, df and dx1 are required");
  return b4r_ffn32w_bwd(d, d->scratch, fpre, df, dx1, records_ready != 0, (hipStream_t)stream);
}
