// Masked-LM head of a TRAIN step without materialised logits, round 4's rebuild of b4r_head_rx.hip on 32 x 32 tiles (hidden size 64;
// the 16-row-tile kernels keep 128 / 256 and serve as the A/B partner: B4R_HEAD32=0).  Same mathematics, same outputs:
//
//   logits x[m,v] = T[m,:].E[v,:] + b[v]      (T = transform output [M,H], E = tied item table [V,H];  bert4rec_model.py:76-81,143)
//   loss_m = logsumexp_v x[m,v] - x[m,y_m] ;  g[m,v] = softmax(x[m,:])[v] - [v == y_m]      (trainer_utils.py:12-23)
//   dT = g.E ;  dE = g^T.T ;  db = column sums of g
//
// What changed against the 16-row-tile kernels (DESIGN.md §4.2 has the measurements):
//   * v_mfma_f32_32x32x16_f16 blocks: half the LDS fragment bytes and a quarter of the LDS instructions per logit;
//   * fp16 hi / lo operand pairs instead of bf16: x = hi + lo with 11 + 11 significant bits.  The logits are the three-term product
//     Th.Eh + Th.El + Tl.Eh (2^-22 relative: tighter than bf16's 2^-16), and the second product of a tile takes its accumulator-born
//     operand -- the probabilities p in [0, 2^8] of the forward, the softmax gradients g in [-1, 1] of dE -- as ONE fp16 value (2^-12
//     relative on every p or g; sums and the loss are formed from the unrounded fp32 values): two matrix instructions per k-step
//     instead of three, and the hi / lo split of 16 values per lane and tile -- 40 of the ~145 vector instructions of a step -- is gone.
//     The vector pipe of a SIMD retires one wave-instruction per ~4 cycles whatever the number of waves (tools/ubench/overlap32.hip),
//     so the instruction count of the softmax IS the budget: a step of two waves hides ~4.5 of them per matrix instruction;
//   * the logit block is formed with the SUMMATION index of the following product on the accumulator's rows (b4r_tile32.h), so p / g go
//     from the accumulator registers straight into the next product as its B operand: nothing but the staged tiles is read from LDS;
//   * the swept operand (E in the forward, T in dE) is converted to fp16 hi / lo panel images ONCE per step by head32_pack_kernel
//     (the 16-row kernels converted every chunk in every workgroup: 80 x the table in the forward) and arrives in an 8-slot LDS ring
//     by LDS-DMA (global_load_lds, 1 KB per wave-instruction, no registers, no vector work) behind counted vmcnt waits: two tiles
//     are in flight while two are read; one barrier per 32-row tile;
//   * the bias (forward) / -lse and the labels (dE) of a tile travel in the tile's record and become the INITIAL VALUE of the logit
//     accumulator: no bias add, and in dE no subtraction in front of the exponential;
//   * software pipeline: a step issues the logit products of tile i + 1, the value products of tile i - 1 and, between them one slice
//     at a time, the vector work of tile i (and the row maximum / argmax bookkeeping of tile i + 1) as ONE hand-ordered stream.
#include "b4r_head32_pack.h"

// H32_PROF (tools/build_variant.sh h32prof b4r_head32.hip -DH32_PROF): lane 0 of waves 0 and 5 of two workgroups stamps the shader clock
#ifdef H32_PROF
__device__ long long g_h32_prof[4][128];
#define H32_MARK(k) do { if ((blockIdx.x == 0 || blockIdx.x == 7) && blockIdx.y == 1 && (threadIdx.x == 0 || threadIdx.x == 320) && (k) < 128) \
    g_h32_prof[(blockIdx.x == 7 ? 2 : 0) + (threadIdx.x ? 1 : 0)][k] = clock64(); } while (0)
#else
#define H32_MARK(k) do { } while (0)
#endif

// timing experiments only (tools/build_variant.sh ... -DH32_EXP=..): 1 no head (maximum / argmax / reference), 2 no vector slices,
// 4 no logit products, 8 no value products, 16 no barrier per step, 32 fragment reads of the first triple only
#ifndef H32_EXP
#define H32_EXP 0
#endif

namespace {

__device__ __forceinline__ f32x16 mfma32h(const f16x8 a, const f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f16x8 h32_row_at(const char* a) { return *reinterpret_cast<const f16x8*>(a); }
__device__ __forceinline__ f16x8 h32_tr_pair(const char* a, const char* b) { return __builtin_bit_cast(f16x8, tr_pair(a, b)); }

constexpr int H32_WAVES = 8;                       // 8 waves x 32 rows = 256 rows of T (forward) / of E (dE) per workgroup
constexpr int H32_ROWS = 32 * H32_WAVES;
constexpr int H32_RING = 16;                       // LDS slots (a power of two): 132 KB at hidden size 64 -- one workgroup per CU anyway
constexpr int H32_AHEAD = 8;                       // an even step i requests tiles i + 8, i + 9: tiles i - 1 .. i + 3 are read, i + 4 .. i + 7 travel
// One wait + barrier per PAIR of steps (a barrier per step cost ~0.2 us of a 1.1 us step: arrival skew of eight waves).  At the barrier of
// an even step i the tiles up to i + 3 have landed: steps i, i + 1 read tiles i + 1, i + 2, and the first fragments of tile i + 3 are
// requested at the end of step i + 1, across the next barrier.

template <int NP>
__global__ __launch_bounds__(256) void head32_pack_kernel(H32PackP p) { h32_pack_tile<NP>(p, (int)blockIdx.x); }

// ---------------------------------------------------------------------------------------------------------------------------
// the tile ring.  Every wave copies its share of a record (NP / 2 pieces of 1 KB), wave 0 also the side block: per tile a wave has
// G (wave 0: G + 1) vector-memory operations in flight, and `vmcnt(K groups)` says "all but the K youngest tiles have landed".
// ---------------------------------------------------------------------------------------------------------------------------
// The copy instruction is written as inline asm: hipcc tracks an LDS-DMA builtin as a pending LDS write and puts `s_waitcnt vmcnt(0)`
// in front of every ds_read whose address it cannot prove distinct from the copy's destination -- with ring slots indexed by the loop
// counter that is every fragment read, i.e. the copies were drained once per tile (seen in the .s of the first version).  The waits
// of this kernel are counted by hand (h32_wait), so the compiler need not know.  M0 = the LDS destination of the wave's 1 KB piece.
__device__ __forceinline__ void h32_dma16(const char* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ void h32_dma4(const char* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}
__device__ __forceinline__ unsigned h32_lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int NP, int WAVES = H32_WAVES>
__device__ __forceinline__ void h32_issue(const char* recs, int64_t tile, char* slot, int wave, int lane) {
  constexpr int G = 4 * NP / WAVES;            // 1 KB pieces of a record per wave
  const char* src = recs + tile * h32_rec(NP);
  const unsigned dst = h32_lds_addr(slot);
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const int piece = wave * G + q;
    h32_dma16(src + piece * 1024 + lane * 16, dst + piece * 1024);
  }
  if (wave == 0) h32_dma4(src + NP * P_TILE + lane * 4, dst + NP * P_TILE);
}
// all but the K youngest tiles of this wave's copies have landed (a compile-time count per wave kind)
template <int NP, int K, int WAVES = H32_WAVES>
__device__ __forceinline__ void h32_wait(int wave) {
  constexpr int G = 4 * NP / WAVES;
  static_assert(K * (G + 1) < 64, "vmcnt is a 6-bit counter");
  if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * (G + 1)) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K * G) : "memory");
}
// (no lgkmcnt wait in front: every fragment read of the finished step has been consumed by its product, and the reads requested for
// the next step target a slot that no copy after this barrier overwrites -- they stay in flight across it)
__device__ __forceinline__ void h32_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

struct H32P {
  const float* own; const char* recs;    // the rows a wave keeps in registers ([*, H] fp32) and the records it sweeps
  int n_own, n_tiles;                    // rows of `own`, tiles of `recs`
  int tiles_per_slice;
  float* part; int M;                    // forward: [slices][M][H + 8] records, then the compact [slices][M][2]
  const float* bias; float* slab; float* bslab; int V;   // dE: [slices][V][H], [slices][V]
};

// the wave's 32 rows of `own` (x log2(e)) as B operands: B[k = 32 p + 16 s + 8 h + j][column r]
template <int NP>
__device__ __forceinline__ void h32_own_rows(const float* own, int64_t row, int h, f16x8 (&oh)[NP][2], f16x8 (&ol)[NP][2]) {
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) h32_split8(load8(own + row * (32 * NP) + 32 * p + 16 * s + 8 * h) * LOG2E, oh[p][s], ol[p][s]);
}
// X^T[row of the tile][column r] = init + tile rows . own^T   (three terms: lo.hi + hi.lo + hi.hi)
template <int NP>
__device__ __forceinline__ f32x16 h32_logits(const char* slot, const Lane32& lk, f32x16 x, const f16x8 (&oh)[NP][2],
                                             const f16x8 (&ol)[NP][2]) {
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char* a = slot + p * P_TILE + lk.rowc[s];
      const f16x8 ah = h32_row_at(a), al = h32_row_at(a + P_IMG);
      x = mfma32h(al, oh[p][s], x);
      x = mfma32h(ah, ol[p][s], x);
      x = mfma32h(ah, oh[p][s], x);
    }
  return x;
}
// acc[p]^T[feature][column r] += tile^T[feature][row] . w[row][column r]   (w = the accumulator registers of the logit block as ONE
// fp16 value each; the tile as hi + lo: two terms)
template <int NP>
__device__ __forceinline__ void h32_feed(const char* slot, const Lane32& lk, const f16x8 (&w)[2], f32x16 (&acc)[NP]) {
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char* a = slot + p * P_TILE;
      acc[p] = mfma32h(h32_tr_pair(a + P_IMG + lk.trp[s][0], a + P_IMG + lk.trp[s][1]), w[s], acc[p]);
      acc[p] = mfma32h(h32_tr_pair(a + lk.trp[s][0], a + lk.trp[s][1]), w[s], acc[p]);
    }
}
// row (t & 3) + 8 (t >> 2) + 4 h of the tile in register t (the D layout's rows)
__device__ __forceinline__ int h32_row_of(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------------------------------------------
// One step's matrix work and vector work as ONE hand-ordered instruction stream.  The waves issue in order: an MFMA that waits for the
// matrix pipe holds back every vector instruction behind it, so products and vector work of a wave only overlap when they ALTERNATE
// in program order -- and hipcc clusters the products in front of the vector work (seen in the .s; a sched_group_barrier pipeline
// is dropped as unsatisfiable because the fragment reads and their address arithmetic sit among the candidates).  So the step is
// written as 20 NP chunks of {one MFMA, one slice of the vector work, the fragment reads of the NEXT product group}, each fenced with
// sched_barrier(0).  The vector work of the 16 logits x of a lane is cut into 16 slices, two per pair (2q, 2q + 1):
//     0: e = exp2(x + c)            1: [HIT: e -= onehot]  sum += e;  w = fp16(e) (packed pair)
// ---------------------------------------------------------------------------------------------------------------------------
struct H32Vec {
  float e[16];
  uint32_t hw[8];
  uint32_t ow[8];   // HIT: -onehot of the pair as packed fp16 (exact: -1 or 0)
};
// slice k (compile-time after unrolling) of the vector work: x = the logit registers, c = the lane's additive constant, sum += e.
// HIT (dE, tiles with a label among the wave's items only): sum -= (yy[t] == v), and w.ow = -onehot as an exact fp16 operand
template <bool HIT>
__device__ __forceinline__ void h32_vslice(int k, const f32x16& x, float c, float& sum, H32Vec& w, const int (&yy)[16], int v) {
  // the empty asm statements pin a slice's results to its place in the stream (pure arithmetic is otherwise sunk to its first use --
  // the end of the step -- before the machine scheduler ever sees the fences)
  const int q = k >> 1;
  if ((k & 1) == 0) {
    w.e[2 * q] = ex2(x[2 * q] + c);
    w.e[2 * q + 1] = ex2(x[2 * q + 1] + c);
    asm volatile("" : "+v"(w.e[2 * q]), "+v"(w.e[2 * q + 1]));
  } else {
    sum += w.e[2 * q];
    sum += w.e[2 * q + 1];
    w.hw[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector((b4r_f32x2){w.e[2 * q], w.e[2 * q + 1]}, f16x2));
    if (HIT) {
      // the label's -1 does NOT go through the fp16 rounding of p (p - 1 rounds to 11 bits: 2.4e-4 of |T| lost on exactly the
      // entries that carry the label's gradient): it travels as its own exact fp16 operand, -1 or 0, into two more products
      const bool c0 = yy[2 * q] == v, c1 = yy[2 * q + 1] == v;
      sum -= c0 ? 1.0f : 0.0f;
      sum -= c1 ? 1.0f : 0.0f;
      w.ow[q] = (c0 ? 0xBC00u : 0u) | (c1 ? 0xBC000000u : 0u);
      asm volatile("" : "+v"(w.ow[q]));
    }
    asm volatile("" : "+v"(sum), "+v"(w.hw[q]));
  }
}
__device__ __forceinline__ void h32_frags_of(const H32Vec& w, f16x8 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
    f[s] = __builtin_bit_cast(f16x8, (b4r_u32x4){w.hw[4 * s], w.hw[4 * s + 1], w.hw[4 * s + 2], w.hw[4 * s + 3]});
}
// L(nxt): xn = xn + tile(nxt) rows . own^T ;  F(prv): acc += tile(prv)^T . q ;  V: the 16 slices on x -> f
// extra(k): more vector work behind matrix instruction k (the forward's row maximum / argmax bookkeeping on xn, which is complete
// two instructions after the last logit product)
template <int NP, bool HIT, typename Extra>
__device__ __forceinline__ void h32_step_block(const char* nxt, const char* prv, const Lane32& lk, const f16x8 (&oh)[NP][2],
                                               const f16x8 (&ol)[NP][2], f32x16& xn, const f16x8 (&q)[2], f32x16 (&acc)[NP],
                                               const f32x16& x, float c, float& sum, f16x8 (&f)[2], const int (&yy)[16], int v,
                                               Extra extra, f16x8 ah, f16x8 al,     // ah, al: the first logit group's fragments (requested by the caller)
                                               const char* cur = nullptr) {         // HIT: the tile of x (its -onehot products are not deferred)
  constexpr int NT = 2 * NP;                 // operand groups (panel, k-step) of L, and of F
  H32Vec w;
  int k = 0;                                 // matrix instruction index; vector slice k is due behind it (16 slices, 10 NT instructions)
  auto slices = [&]() __attribute__((always_inline)) {
    if (!(H32_EXP & 2) && k < 16) h32_vslice<HIT>(k, x, c, sum, w, yy, v);
    extra(k);
    ++k;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_l = [&](int j, f16x8& ah, f16x8& al) __attribute__((always_inline)) {
    const char* a = nxt + (j >> 1) * P_TILE + lk.rowc[j & 1];
    ah = h32_row_at(a); al = h32_row_at(a + P_IMG);
  };
  auto load_f = [&](int j, f16x8& ah, f16x8& al) __attribute__((always_inline)) {
    const char* a = prv + (j >> 1) * P_TILE;
    const int s = j & 1;
    ah = h32_tr_pair(a + lk.trp[s][0], a + lk.trp[s][1]);
    al = h32_tr_pair(a + P_IMG + lk.trp[s][0], a + P_IMG + lk.trp[s][1]);
  };
  f16x8 nh, nl;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!(H32_EXP & 32)) { if (j + 1 < NT) load_l(j + 1, nh, nl); else load_f(0, nh, nl); }   // the next group's fragments travel during this one
    if (!(H32_EXP & 4)) xn = mfma32h(al, oh[j >> 1][j & 1], xn);
    slices();
    if (!(H32_EXP & 4)) xn = mfma32h(ah, ol[j >> 1][j & 1], xn);
    slices();
    if (!(H32_EXP & 4)) xn = mfma32h(ah, oh[j >> 1][j & 1], xn);
    slices();
    if (!(H32_EXP & 32)) { ah = nh; al = nl; }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!(H32_EXP & 32)) { if (j + 1 < NT) load_f(j + 1, nh, nl); }
    if (!(H32_EXP & 8)) acc[j >> 1] = mfma32h(al, q[j & 1], acc[j >> 1]);
    slices();
    if (!(H32_EXP & 8)) acc[j >> 1] = mfma32h(ah, q[j & 1], acc[j >> 1]);
    slices();
    if (!(H32_EXP & 32)) { ah = nh; al = nl; }
  }
  if (H32_EXP & 2) {
#pragma unroll
    for (int qq = 0; qq < 8; ++qq) w.hw[qq] = __builtin_bit_cast(uint32_t, x[qq] + c);
  }
  h32_frags_of(w, f);
  if (HIT) {   // acc += tile(cur)^T . (-onehot): exact, hi and lo image
    f16x8 of[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
      of[s] = __builtin_bit_cast(f16x8, (b4r_u32x4){w.ow[4 * s], w.ow[4 * s + 1], w.ow[4 * s + 2], w.ow[4 * s + 3]});
    h32_feed<NP>(cur, lk, of, acc);
  }
}
__device__ __forceinline__ void h32_zero_frags(f16x8 (&a)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[s][e] = (_Float16)0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward: grid (blocks of 256 rows of T, V slices); wave = 32 rows of T x the slice's tiles of E.
// Software pipeline over the tiles, one step = one basic block of 24 matrix instructions with the step's vector work between them:
//   L(i + 1)  the logits of tile i + 1                      (12 MFMA, row reads of slot i + 1)
//   F(i - 1)  acc += E^T . p of tile i - 1                   (12 MFMA, transposed reads of slot i - 1; p(i - 1) waits in registers)
//   V(i)      exp2 / row sum / hi-lo split of tile i's logits -> p(i)
// ---------------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64 * H32_WAVES, 2) void head32_fwd_kernel(H32P p) {
  extern __shared__ __attribute__((aligned(16))) char smem_h32[];
  constexpr int H = 32 * NP, REC = h32_rec(NP), PART_LD = part_ld(NP);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane32 lk = lane32(lane);
  const int r = lk.r, h = lk.h;
  const int m = blockIdx.x * H32_ROWS + 32 * wave + r;
  f16x8 th[NP][2], tl[NP][2];
  H32_MARK(0);
  h32_own_rows<NP>(p.own, min(m, p.M - 1), h, th, tl);
  __builtin_amdgcn_sched_barrier(0);
  H32_MARK(1);
  const int t0 = blockIdx.y * p.tiles_per_slice;
  const int n = min(p.tiles_per_slice, p.n_tiles - t0);                 // >= 1
  auto slot_of = [&](int i) __attribute__((always_inline)) { return smem_h32 + (i & (H32_RING - 1)) * REC; };
  // tiles beyond the slice are copies of its last tile: every step has the same shape, the surplus logits are never used
  auto issue = [&](int i) __attribute__((always_inline)) { h32_issue<NP>(p.recs, t0 + min(i, n - 1), slot_of(i), wave, lane); };
#pragma unroll
  for (int q = 0; q < H32_AHEAD; ++q) issue(q);

  float mx = -INFINITY, sum = 0.f, best = -INFINITY;   // log2 units
  int bidx = 0x7fffffff;
  f32x16 acc[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) acc[pp] = zero16();

  const int no_labels[16] = {};
  H32_MARK(2);
  h32_wait<NP, H32_AHEAD - 2>(wave);                         // tiles 0, 1
  h32_barrier();
  H32_MARK(3);
  // Row maximum and argmax of a logit block X (tile index it): `best` = the lane's largest logit so far, `rec` = the 16 logits of the
  // tile it was found in, `btile` that tile -- the POSITION inside the tile is only looked up once, after the sweep (a position search
  // per step was a third of the step's vector work).  Returns the block's maximum over the lane's 16 columns.
  float rec[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) rec[t] = -INFINITY;
  int btile = -1;
  // `mx` only has to be COMMON to the two lanes of a row and close enough to the row maximum that 2^(x - mx) cannot overflow (the
  // reasoning of b4r_head_rx.hip): it moves -- one exchange, one rescale -- only when some logit of the wave exceeds its row's reference
  // by more than 2^SLACK: at the first tile of a slice (mx = -inf) and then almost never.  Everything that is still at the old reference
  // is brought along exactly once: the pending products (fragments q of the tile in `slot`) are finished first, then acc and sum
  // rescaled.
  auto move_reference = [&](float pl8, const char* slot, f16x8 (&q)[2]) __attribute__((always_inline)) {
    constexpr float SLACK = 8.0f;
    if (__builtin_amdgcn_ballot_w64(pl8 > mx + SLACK) != 0) {
      h32_feed<NP>(slot, lk, q, acc);
      h32_zero_frags(q);
      const float pm = fmaxf(pl8, other_half(pl8, h));
      const float mnew = fmaxf(mx, pm);                      // finite: the first tile of a slice holds real columns
      const float alpha = (mx == mnew) ? 1.0f : ex2(mx - mnew);
      sum *= alpha;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) acc[pp] = acc[pp] * alpha;
      mx = mnew;
    }
  };
  // One step (entry: S = the logits of tile i, already examined: mx is valid for them; q = the probabilities of tile i - 1):
  //   matrix instructions 0 .. 6 NP - 1:  L(i + 1) -> Sn          6 NP .. 10 NP - 1:  F(i - 1): acc += E^T . p(i - 1)
  //   vector slices, one behind every matrix instruction: V(i) -> ph / pl; behind the products 6 NP + 2 ..: the maximum of Sn and the
  //   argmax bookkeeping.  Then the (rare) move of the reference for tile i + 1.
  // what a step's first matrix instruction waits for, requested at the end of the step before it (and in flight across the barrier)
  f32x16 pre_init;
  f16x8 pre_h, pre_l;
  auto request = [&](int t) __attribute__((always_inline)) {
    const char* sl = slot_of(t);
    pre_init = rows_of(reinterpret_cast<const float*>(sl + NP * P_TILE), h);
    pre_h = h32_row_at(sl + lk.rowc[0]);
    pre_l = h32_row_at(sl + P_IMG + lk.rowc[0]);
  };
  auto step = [&](int i, bool sync, f32x16& S, f32x16& Sn, f16x8 (&q)[2], f16x8 (&pw)[2]) __attribute__((always_inline)) {
    const char* prv = slot_of(i > 0 ? i - 1 : 0);            // (step 0: q is zero)
    const char* nxt = slot_of(i + 1);
    H32_MARK(4 + 4 * i);
    if (sync) {
      h32_wait<NP, 4>(wave);                                 // the tiles up to i + 3 have landed (this wave's pieces) ...
      H32_MARK(5 + 4 * i);
      if (!(H32_EXP & 16)) h32_barrier();                    // ... every wave's; and every wave is done with the slots requested next
      H32_MARK(6 + 4 * i);
      issue(i + H32_AHEAD);
      issue(i + H32_AHEAD + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    H32_MARK(7 + 4 * i);
    Sn = pre_init;                                           // the bias of the tile's rows as the accumulator's start (requested a step ago)
    float m5[5], pl8 = 0.f;
    bool newrec = false;
    constexpr int K0 = 6 * NP + 2;                           // Sn is complete two matrix instructions after its last product
    auto extra = [&](int k) __attribute__((always_inline)) {
      if (H32_EXP & 1) return;
      const int e = k - K0;
      if (e == 0) {                                          // 3-way maxima of columns 0 .. 11
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          m5[u] = fmaxf(fmaxf(Sn[3 * u], Sn[3 * u + 1]), Sn[3 * u + 2]);
          asm volatile("" : "+v"(m5[u]));
        }
      } else if (e == 1) {
        m5[4] = fmaxf(fmaxf(Sn[12], Sn[13]), Sn[14]);
        m5[0] = fmaxf(fmaxf(m5[0], m5[1]), m5[2]);
        pl8 = fmaxf(fmaxf(fmaxf(m5[3], m5[4]), Sn[15]), m5[0]);
        newrec = (i + 1 < n) && pl8 > best;                  // (tiles beyond the slice are copies of its last tile)
        best = newrec ? pl8 : best;
        btile = newrec ? i + 1 : btile;
        asm volatile("" : "+v"(best), "+v"(btile), "+v"(pl8));
      } else if (e >= 2 && e < 6) {                          // the record's logits: four per slice
#pragma unroll
        for (int t = 4 * (e - 2); t < 4 * (e - 2) + 4; ++t) {
          rec[t] = newrec ? Sn[t] : rec[t];
          asm volatile("" : "+v"(rec[t]));
        }
      }
    };
    static_assert(K0 + 6 <= 10 * NP, "the bookkeeping slices must fit behind the value products");
    h32_step_block<NP, false>(nxt, prv, lk, th, tl, Sn, q, acc, S, -mx, sum, pw, no_labels, 0, extra, pre_h, pre_l);
    request(i + 2);                                          // the first fragments and the bias of the NEXT step's logit tile
    if (!(H32_EXP & 1)) move_reference(pl8, slot_of(i), pw);   // pending now: the probabilities of tile i, just formed (old reference)
  };
  f32x16 S0 = h32_logits<NP>(slot_of(0), lk, rows_of(reinterpret_cast<const float*>(slot_of(0) + NP * P_TILE), h), th, tl), S1;
  f16x8 PA[2], PB[2];                                       // the two probability-fragment states: (previous, current) alternate
  h32_zero_frags(PA);
  h32_zero_frags(PB);
  {   // tile 0: its maximum, the first reference (mx = -inf: always moved here), the first record
    float pl8 = fmaxf(fmaxf(S0[0], S0[1]), fmaxf(S0[2], S0[3]));
#pragma unroll
    for (int t = 4; t < 16; t += 4) pl8 = fmaxf(pl8, fmaxf(fmaxf(S0[t], S0[t + 1]), fmaxf(S0[t + 2], S0[t + 3])));
    if (pl8 > best) {
      best = pl8; btile = 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) rec[t] = S0[t];
    }
    move_reference(pl8, slot_of(0), PA);
  }
  request(1);
  for (int i = 0; i < n; i += 2) {
    step(i, true, S0, S1, PA, PB);
    if (i + 1 < n) step(i + 1, false, S1, S0, PB, PA);
  }
  H32_MARK(120);
  if (n & 1) h32_feed<NP>(slot_of(n - 1), lk, PB, acc);      // the products of the last tile
  else h32_feed<NP>(slot_of(n - 1), lk, PA, acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the surplus copies
  if (btile >= 0) {                                          // the position of the lane's best logit inside its tile: lowest = lowest column
    int j = 15;
#pragma unroll
    for (int jj = 14; jj >= 0; --jj) j = (rec[jj] == best) ? jj : j;
    bidx = 32 * (t0 + btile) + h32_row_of(j, h);
  }
  H32_MARK(121);
  // the two lanes of a row hold disjoint columns: combine
  sum += other_half(sum, h);
  {
    const float ov = other_half(best, h);
    const int oi = (int)other_half_u((unsigned)bidx, h);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (m < p.M) {
    float* dst = p.part + ((int64_t)blockIdx.y * p.M + m) * PART_LD;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(dst + 32 * pp + 8 * g4 + 4 * h) =
            (f32x4){acc[pp][4 * g4], acc[pp][4 * g4 + 1], acc[pp][4 * g4 + 2], acc[pp][4 * g4 + 3]};
    if (h == 0) {
      dst[H] = mx; dst[H + 1] = sum; dst[H + 2] = best; dst[H + 3] = __int_as_float(bidx);
      float* ms = p.part + (int64_t)gridDim.y * p.M * PART_LD + ((int64_t)blockIdx.y * p.M + m) * 2;   // the compact (max, sum) copy
      ms[0] = mx; ms[1] = sum;
    }
  }
  H32_MARK(122);
}

// ---------------------------------------------------------------------------------------------------------------------------
// dE / db: grid (blocks of 256 rows of E, M slices); wave = 32 rows of E x the slice's tiles of T.  The same pipeline:
//   L(i + 1)  X = b[v] - lse[m] + T(i + 1) . E^T            F(i - 1)  dE^T += T^T . g of tile i - 1
//   V(i)      g = exp2(X) - [y_m == v], column sums, hi-lo split
// ---------------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(64 * H32_WAVES, 2) void head32_dE_kernel(H32P p) {
  extern __shared__ __attribute__((aligned(16))) char smem_h32[];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr int H = 32 * NP, REC = h32_rec(NP);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane32 lk = lane32(lane);
  const int r = lk.r, h = lk.h;
  const int v = blockIdx.x * H32_ROWS + 32 * wave + r;
  const bool vlive = v < p.V;
  f16x8 eh[NP][2], el[NP][2];
  h32_own_rows<NP>(p.own, min(v, p.V - 1), h, eh, el);
  const float bv = vlive ? p.bias[v] * LOG2E : -INFINITY;    // -inf => zero probability, and no label equals v >= V
  __builtin_amdgcn_sched_barrier(0);
  const int t0 = blockIdx.y * p.tiles_per_slice;
  const int n = min(p.tiles_per_slice, p.n_tiles - t0);
  auto slot_of = [&](int i) __attribute__((always_inline)) { return smem_h32 + (i & (H32_RING - 1)) * REC; };
  auto issue = [&](int i) __attribute__((always_inline)) { h32_issue<NP>(p.recs, t0 + min(i, n - 1), slot_of(i), wave, lane); };
#pragma unroll
  for (int q = 0; q < H32_AHEAD; ++q) issue(q);

  float dbsum = 0.f;
  const int no_labels[16] = {};
  f32x16 acc[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) acc[pp] = zero16();
  // X[row m of the tile][column v] starts from -lse[m] (log2 units, the tile's side block); b[v] is added in front of the exponential
  auto x_init = [&](const char* slot) __attribute__((always_inline)) -> f32x16 {
    return rows_of(reinterpret_cast<const float*>(slot + NP * P_TILE), h);
  };
  const int v0 = blockIdx.x * H32_ROWS + 32 * wave;          // the wave's first item
  h32_wait<NP, H32_AHEAD - 2>(wave);                         // tiles 0, 1
  h32_barrier();
  f32x16 pre_init;
  f16x8 pre_h, pre_l;
  auto request = [&](int t) __attribute__((always_inline)) {
    const char* sl = slot_of(t);
    pre_init = x_init(sl);
    pre_h = h32_row_at(sl + lk.rowc[0]);
    pre_l = h32_row_at(sl + P_IMG + lk.rowc[0]);
  };
  auto step = [&](int i, bool sync, f32x16& X, f32x16& Xn, f16x8 (&q)[2], f16x8 (&gw)[2]) __attribute__((always_inline)) {
    if (sync) {
      h32_wait<NP, 4>(wave);
      h32_barrier();
      issue(i + H32_AHEAD);
      issue(i + H32_AHEAD + 1);
    }
    const char* cur = slot_of(i);
    const char* prv = slot_of(i > 0 ? i - 1 : 0);
    const char* nxt = slot_of(i + 1);
    // does a label of this tile's 32 rows name one of the wave's 32 items?  (one label per lane, a wave-uniform answer: mostly no)
    const int* ly = reinterpret_cast<const int*>(cur + NP * P_TILE) + 32;
    const bool hit = __builtin_amdgcn_ballot_w64((unsigned)(ly[r] - v0) < 32u) != 0;
    __builtin_amdgcn_sched_barrier(0);
    Xn = pre_init;
    if (hit) {
      int yy[16];
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {
        const i32x4 y4 = *reinterpret_cast<const i32x4*>(ly + 8 * gp + 4 * h);   // the labels of rows 8 gp + 4 h .. + 3
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) yy[4 * gp + e4] = y4[e4];
      }
      h32_step_block<NP, true>(nxt, prv, lk, eh, el, Xn, q, acc, X, bv, dbsum, gw, yy, v, [](int) {}, pre_h, pre_l, cur);
    } else {
      h32_step_block<NP, false>(nxt, prv, lk, eh, el, Xn, q, acc, X, bv, dbsum, gw, no_labels, v, [](int) {}, pre_h, pre_l);
    }
    request(i + 2);
  };
  f32x16 X0 = h32_logits<NP>(slot_of(0), lk, x_init(slot_of(0)), eh, el), X1;
  f16x8 GA[2], GB[2];
  h32_zero_frags(GA);
  h32_zero_frags(GB);
  request(1);
  for (int i = 0; i < n; i += 2) {
    step(i, true, X0, X1, GA, GB);
    if (i + 1 < n) step(i + 1, false, X1, X0, GB, GA);
  }
  if (n & 1) h32_feed<NP>(slot_of(n - 1), lk, GB, acc);
  else h32_feed<NP>(slot_of(n - 1), lk, GA, acc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dbsum += other_half(dbsum, h);
  if (vlive) {
    float* dst = p.slab + ((int64_t)blockIdx.y * p.V + v) * H;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(dst + 32 * pp + 8 * g4 + 4 * h) =
            (f32x4){acc[pp][4 * g4], acc[pp][4 * g4 + 1], acc[pp][4 * g4 + 2], acc[pp][4 * g4 + 3]};
    if (h == 0) p.bslab[(int64_t)blockIdx.y * p.V + v] = dbsum;
  }
}

// ===========================================================================================================================
// hidden size 128 / 256 (NP = 4 / 8 panels).  A tile now carries 6 NP + 4 NP = 40 / 80 matrix instructions, so the vector work of tile
// i fits between the logit products of tile i + 1 and the value products can take tile i's own probabilities IN THE SAME STEP: no
// pending operand, two resident tiles (i, i + 1) + two in flight in a 4-slot ring, one barrier per step (a step is 1.3 - 2.6 k cycles of
// matrix work per wave).  NP = 4: eight waves of 32 rows, two per SIMD; NP = 8: the T rows (128 registers) and the accumulators (128)
// of 32 rows need the whole register file: FOUR waves of 32 rows, one per SIMD, 512 registers each -- at 1.25 vector instructions per
// matrix instruction a single wave keeps the matrix pipe busy (tools/ubench/overlap32.hip: up to 4 hide completely).
// ===========================================================================================================================
constexpr int H32W_RING = 4;
template <int NP, bool HIT, typename Extra>
__device__ __forceinline__ void h32_step_block_nd(const char* nxt, const char* cur, const Lane32& lk, const f16x8 (&oh)[NP][2],
                                                  const f16x8 (&ol)[NP][2], f32x16& xn, f32x16 (&acc)[NP], const f32x16& x, float c,
                                                  float& sum, const int (&yy)[16], int v, Extra extra) {
  constexpr int NT = 2 * NP;
  H32Vec w;
  int k = 0;
  auto slices = [&]() __attribute__((always_inline)) {
    if (k < 16) h32_vslice<HIT>(k, x, c, sum, w, yy, v);
    extra(k);
    ++k;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto load_l = [&](int j, f16x8& ah, f16x8& al) __attribute__((always_inline)) {
    const char* a = nxt + (j >> 1) * P_TILE + lk.rowc[j & 1];
    ah = h32_row_at(a); al = h32_row_at(a + P_IMG);
  };
  auto load_f = [&](int j, f16x8& ah, f16x8& al) __attribute__((always_inline)) {
    const char* a = cur + (j >> 1) * P_TILE;
    const int s = j & 1;
    ah = h32_tr_pair(a + lk.trp[s][0], a + lk.trp[s][1]);
    al = h32_tr_pair(a + P_IMG + lk.trp[s][0], a + P_IMG + lk.trp[s][1]);
  };
  f16x8 ah, al, nh, nl;
  load_l(0, ah, al);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j + 1 < NT) load_l(j + 1, nh, nl); else load_f(0, nh, nl);
    xn = mfma32h(al, oh[j >> 1][j & 1], xn); slices();
    xn = mfma32h(ah, ol[j >> 1][j & 1], xn); slices();
    xn = mfma32h(ah, oh[j >> 1][j & 1], xn); slices();
    ah = nh; al = nl;
  }
  static_assert(6 * NP >= 16, "the vector slices must be finished when the value products start");
  f16x8 f[2], of[2];
  h32_frags_of(w, f);
#pragma unroll
  for (int s = 0; s < 2; ++s)
    of[s] = HIT ? __builtin_bit_cast(f16x8, (b4r_u32x4){w.ow[4 * s], w.ow[4 * s + 1], w.ow[4 * s + 2], w.ow[4 * s + 3]}) : f[s];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j + 1 < NT) load_f(j + 1, nh, nl);
    acc[j >> 1] = mfma32h(al, f[j & 1], acc[j >> 1]); slices();
    acc[j >> 1] = mfma32h(ah, f[j & 1], acc[j >> 1]); slices();
    if (HIT) {   // the labels' exact -1 (see h32_vslice): two more products on the same fragments
      acc[j >> 1] = mfma32h(al, of[j & 1], acc[j >> 1]);
      acc[j >> 1] = mfma32h(ah, of[j & 1], acc[j >> 1]);
    }
    ah = nh; al = nl;
  }
}

template <int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void head32w_fwd_kernel(H32P p) {
  extern __shared__ __attribute__((aligned(16))) char smem_h32[];
  constexpr int H = 32 * NP, REC = h32_rec(NP), PART_LD = part_ld(NP), ROWS = 32 * WAVES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane32 lk = lane32(lane);
  const int r = lk.r, h = lk.h;
  const int m = blockIdx.x * ROWS + 32 * wave + r;
  f16x8 th[NP][2], tl[NP][2];
  h32_own_rows<NP>(p.own, min(m, p.M - 1), h, th, tl);
  __builtin_amdgcn_sched_barrier(0);
  const int t0 = blockIdx.y * p.tiles_per_slice;
  const int n = min(p.tiles_per_slice, p.n_tiles - t0);                 // >= 1
  auto slot_of = [&](int i) __attribute__((always_inline)) { return smem_h32 + (i & (H32W_RING - 1)) * REC; };
  auto issue = [&](int i) __attribute__((always_inline)) { h32_issue<NP, WAVES>(p.recs, t0 + min(i, n - 1), slot_of(i), wave, lane); };
  issue(0); issue(1); issue(2);

  float mx = -INFINITY, sum = 0.f, best = -INFINITY;   // log2 units
  int bidx = 0x7fffffff, btile = -1;
  float rec[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) rec[t] = -INFINITY;
  f32x16 acc[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) acc[pp] = zero16();
  const int no_labels[16] = {};
  // the reference of the running sums (b4r_head_rx.hip's reasoning): moved when a logit exceeds it by more than 2^SLACK; nothing is
  // pending between two steps here, so only acc and sum are brought along
  auto move_reference = [&](float pl8) __attribute__((always_inline)) {
    constexpr float SLACK = 8.0f;
    if (__builtin_amdgcn_ballot_w64(pl8 > mx + SLACK) != 0) {
      const float pm = fmaxf(pl8, other_half(pl8, h));
      const float mnew = fmaxf(mx, pm);
      const float alpha = (mx == mnew) ? 1.0f : ex2(mx - mnew);
      sum *= alpha;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) acc[pp] = acc[pp] * alpha;
      mx = mnew;
    }
  };
  h32_wait<NP, 2, WAVES>(wave);                              // tile 0
  h32_barrier();
  auto step = [&](int i, f32x16& S, f32x16& Sn) __attribute__((always_inline)) {
    h32_wait<NP, 1, WAVES>(wave);                            // tile i + 1 has landed (this wave's pieces) ...
    h32_barrier();                                           // ... every wave's; and every wave is done with tile i - 1's slot
    issue(i + 3);
    const char* cur = slot_of(i);
    const char* nxt = slot_of(i + 1);
    __builtin_amdgcn_sched_barrier(0);
    Sn = rows_of(reinterpret_cast<const float*>(nxt + NP * P_TILE), h);
    float m5[5], pl8 = 0.f;
    bool newrec = false;
    constexpr int K0 = 6 * NP + 2;
    auto extra = [&](int k) __attribute__((always_inline)) {
      const int e = k - K0;
      if (e == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          m5[u] = fmaxf(fmaxf(Sn[3 * u], Sn[3 * u + 1]), Sn[3 * u + 2]);
          asm volatile("" : "+v"(m5[u]));
        }
      } else if (e == 1) {
        m5[4] = fmaxf(fmaxf(Sn[12], Sn[13]), Sn[14]);
        m5[0] = fmaxf(fmaxf(m5[0], m5[1]), m5[2]);
        pl8 = fmaxf(fmaxf(fmaxf(m5[3], m5[4]), Sn[15]), m5[0]);
        newrec = (i + 1 < n) && pl8 > best;
        best = newrec ? pl8 : best;
        btile = newrec ? i + 1 : btile;
        asm volatile("" : "+v"(best), "+v"(btile), "+v"(pl8));
      } else if (e >= 2 && e < 6) {
#pragma unroll
        for (int t = 4 * (e - 2); t < 4 * (e - 2) + 4; ++t) {
          rec[t] = newrec ? Sn[t] : rec[t];
          asm volatile("" : "+v"(rec[t]));
        }
      }
    };
    h32_step_block_nd<NP, false>(nxt, cur, lk, th, tl, Sn, acc, S, -mx, sum, no_labels, 0, extra);
    move_reference(pl8);
  };
  f32x16 S0 = h32_logits<NP>(slot_of(0), lk, rows_of(reinterpret_cast<const float*>(slot_of(0) + NP * P_TILE), h), th, tl), S1;
  {
    float pl8 = fmaxf(fmaxf(S0[0], S0[1]), fmaxf(S0[2], S0[3]));
#pragma unroll
    for (int t = 4; t < 16; t += 4) pl8 = fmaxf(pl8, fmaxf(fmaxf(S0[t], S0[t + 1]), fmaxf(S0[t + 2], S0[t + 3])));
    if (pl8 > best) {
      best = pl8; btile = 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) rec[t] = S0[t];
    }
    move_reference(pl8);
  }
  for (int i = 0; i < n; i += 2) {
    step(i, S0, S1);
    if (i + 1 < n) step(i + 1, S1, S0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (btile >= 0) {
    int j = 15;
#pragma unroll
    for (int jj = 14; jj >= 0; --jj) j = (rec[jj] == best) ? jj : j;
    bidx = 32 * (t0 + btile) + h32_row_of(j, h);
  }
  sum += other_half(sum, h);
  {
    const float ov = other_half(best, h);
    const int oi = (int)other_half_u((unsigned)bidx, h);
    if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
  }
  if (m < p.M) {
    float* dst = p.part + ((int64_t)blockIdx.y * p.M + m) * PART_LD;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(dst + 32 * pp + 8 * g4 + 4 * h) =
            (f32x4){acc[pp][4 * g4], acc[pp][4 * g4 + 1], acc[pp][4 * g4 + 2], acc[pp][4 * g4 + 3]};
    if (h == 0) {
      dst[H] = mx; dst[H + 1] = sum; dst[H + 2] = best; dst[H + 3] = __int_as_float(bidx);
      float* ms = p.part + (int64_t)gridDim.y * p.M * PART_LD + ((int64_t)blockIdx.y * p.M + m) * 2;
      ms[0] = mx; ms[1] = sum;
    }
  }
}

template <int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void head32w_dE_kernel(H32P p) {
  extern __shared__ __attribute__((aligned(16))) char smem_h32[];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr int H = 32 * NP, REC = h32_rec(NP), ROWS = 32 * WAVES;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Lane32 lk = lane32(lane);
  const int r = lk.r, h = lk.h;
  const int v0 = blockIdx.x * ROWS + 32 * wave, v = v0 + r;
  const bool vlive = v < p.V;
  f16x8 eh[NP][2], el[NP][2];
  h32_own_rows<NP>(p.own, min(v, p.V - 1), h, eh, el);
  const float bv = vlive ? p.bias[v] * LOG2E : -INFINITY;
  __builtin_amdgcn_sched_barrier(0);
  const int t0 = blockIdx.y * p.tiles_per_slice;
  const int n = min(p.tiles_per_slice, p.n_tiles - t0);
  auto slot_of = [&](int i) __attribute__((always_inline)) { return smem_h32 + (i & (H32W_RING - 1)) * REC; };
  auto issue = [&](int i) __attribute__((always_inline)) { h32_issue<NP, WAVES>(p.recs, t0 + min(i, n - 1), slot_of(i), wave, lane); };
  issue(0); issue(1); issue(2);
  float dbsum = 0.f;
  const int no_labels[16] = {};
  f32x16 acc[NP];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) acc[pp] = zero16();
  h32_wait<NP, 2, WAVES>(wave);
  h32_barrier();
  auto step = [&](int i, f32x16& X, f32x16& Xn) __attribute__((always_inline)) {
    h32_wait<NP, 1, WAVES>(wave);
    h32_barrier();
    issue(i + 3);
    const char* cur = slot_of(i);
    const char* nxt = slot_of(i + 1);
    const int* ly = reinterpret_cast<const int*>(cur + NP * P_TILE) + 32;
    const bool hit = __builtin_amdgcn_ballot_w64((unsigned)(ly[r] - v0) < 32u) != 0;
    __builtin_amdgcn_sched_barrier(0);
    Xn = rows_of(reinterpret_cast<const float*>(nxt + NP * P_TILE), h);   // -lse of the tile's rows
    if (hit) {
      int yy[16];
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {
        const i32x4 y4 = *reinterpret_cast<const i32x4*>(ly + 8 * gp + 4 * h);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) yy[4 * gp + e4] = y4[e4];
      }
      h32_step_block_nd<NP, true>(nxt, cur, lk, eh, el, Xn, acc, X, bv, dbsum, yy, v, [](int) {});
    } else {
      h32_step_block_nd<NP, false>(nxt, cur, lk, eh, el, Xn, acc, X, bv, dbsum, no_labels, v, [](int) {});
    }
  };
  f32x16 X0 = h32_logits<NP>(slot_of(0), lk, rows_of(reinterpret_cast<const float*>(slot_of(0) + NP * P_TILE), h), eh, el), X1;
  for (int i = 0; i < n; i += 2) {
    step(i, X0, X1);
    if (i + 1 < n) step(i + 1, X1, X0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dbsum += other_half(dbsum, h);
  if (vlive) {
    float* dst = p.slab + ((int64_t)blockIdx.y * p.V + v) * H;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<f32x4*>(dst + 32 * pp + 8 * g4 + 4 * h) =
            (f32x4){acc[pp][4 * g4], acc[pp][4 * g4 + 1], acc[pp][4 * g4 + 2], acc[pp][4 * g4 + 3]};
    if (h == 0) p.bslab[(int64_t)blockIdx.y * p.V + v] = dbsum;
  }
}

int h32_target_wgs() {
  static const int t = getenv("B4R_HEAD32_WGS") ? atoi(getenv("B4R_HEAD32_WGS")) : 256;
  return t < 1 ? 1 : t;
}
// slices of the swept dimension for `own_rows` rows held in registers: about one workgroup per CU
// rows of `own` per workgroup: 8 waves of 32 (hidden 64 / 128), 4 waves of 32 at hidden 256 (one wave per SIMD, 512 registers)
// (dE at hidden 128 as well: its two code versions -- with and without a label among the wave's items -- spill under the 256-register cap
// of two waves per SIMD)
int h32_rows_wg(int H, bool fwd) { return (H == 256 || (H == 128 && !fwd)) ? 128 : 256; }
int h32_slices(int own_rows, int swept_rows, int max_slices, int H, bool fwd) {
  const int blocks = b4r_cdiv(own_rows, h32_rows_wg(H, fwd)), tiles = b4r_cdiv(swept_rows, 32);
  int s = h32_target_wgs() / blocks;                           // never more workgroups than CUs: a workgroup that has to wait for a CU doubles the kernel
  s = s < 1 ? 1 : (s > max_slices ? max_slices : s);
  s = s > tiles ? tiles : s;
  const int per = b4r_cdiv(tiles, s);
  return b4r_cdiv(tiles, per);                                 // no empty slice
}
int64_t h32_rec_floats(int rows, int H) { return (int64_t)b4r_cdiv(rows, 32) * h32_rec(H / 32) / 4; }
int64_t up4l(int64_t x) { return (x + 3) & ~(int64_t)3; }

template <int NP>
int h32_pack(const H32PackP& p, hipStream_t stream) {
  hipLaunchKernelGGL(head32_pack_kernel<NP>, dim3(b4r_cdiv(p.R, 32)), dim3(256), 0, stream, p);
  B4R_CHECK_LAUNCH(p.mode == 0 ? "masked-LM head: item-table images" : "masked-LM head: transform-row images");
  return B4R_OK;
}

}  // namespace

#ifdef H32_PROF
extern "C" int b4r_debug_h32_prof(long long* host_out) {   // the stamps of the last forward launch (after a device synchronisation)
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_h32_prof), 4 * 128 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
#endif

// ---- host side (called by b4r_head_rx.hip's entry points when the 32 x 32 kernels serve the shape) --------------------------
bool b4r_head32_active(int H) {
  static const bool off = getenv("B4R_HEAD32") && atoi(getenv("B4R_HEAD32")) == 0;
  static const bool wide_off = getenv("B4R_HEAD32_WIDE") && atoi(getenv("B4R_HEAD32_WIDE")) == 0;
  return !off && (H == 64 || (!wide_off && (H == 128 || H == 256)));
}
int b4r_head32_fwd_slices(int M, int V, int H) { return h32_slices(M, V, 16, H, true); }
// (at most 16 slabs: the ordered reduction then sums a gradient element in one thread with every load in flight, b4r_gemm.hip)
int b4r_head32_dE_slices(int M, int V, int H) { return h32_slices(V, M, 16, H, false); }
// forward scratch: [slices][M][H + 8] records | [slices][M][2] | the item table's tile records
int64_t b4r_head32_fwd_scratch_floats(int M, int V, int H) {
  return up4l((int64_t)b4r_head32_fwd_slices(M, V, H) * M * (H + 8 + 2)) + h32_rec_floats(V, H);
}
// dE scratch: [slices][V][H] | [slices][V] | the transform rows' tile records
int64_t b4r_head32_dE_scratch_floats(int M, int V, int H) {
  return up4l((int64_t)b4r_head32_dE_slices(M, V, H) * ((int64_t)V * H + V)) + h32_rec_floats(M, H);
}

namespace {
template <int NP, int WAVES, bool FWD>
int h32_launch_sweep(const H32PackP& pk, const H32P& p, int own_rows, int slices, hipStream_t stream) {
  int rc = pk.src != nullptr ? h32_pack<NP>(pk, stream) : B4R_OK;   // (src == NULL: another launch has formed the records)
  if (rc) return rc;
  const bool narrow = NP == 2;
  const size_t lds = (size_t)(narrow ? H32_RING : H32W_RING) * h32_rec(NP);
  const dim3 grid(b4r_cdiv(own_rows, 32 * WAVES), slices), block(64 * WAVES);
  if constexpr (NP == 2) {
    const void* k = FWD ? (const void*)head32_fwd_kernel<2> : (const void*)head32_dE_kernel<2>;
    rc = b4r_raise_lds(k, lds, "masked-LM head sweep");
    if (rc) return rc;
    if (FWD) hipLaunchKernelGGL(head32_fwd_kernel<2>, grid, block, lds, stream, p);
    else hipLaunchKernelGGL(head32_dE_kernel<2>, grid, block, lds, stream, p);
  } else {
    if constexpr (FWD) {
      rc = b4r_raise_lds((const void*)head32w_fwd_kernel<NP, WAVES>, lds, "masked-LM head sweep");
      if (rc) return rc;
      hipLaunchKernelGGL((head32w_fwd_kernel<NP, WAVES>), grid, block, lds, stream, p);
    } else {
      rc = b4r_raise_lds((const void*)head32w_dE_kernel<NP, WAVES>, lds, "masked-LM head sweep");
      if (rc) return rc;
      hipLaunchKernelGGL((head32w_dE_kernel<NP, WAVES>), grid, block, lds, stream, p);
    }
  }
  return B4R_OK;
}
template <bool FWD>
int h32_dispatch(int H, const H32PackP& pk, const H32P& p, int own_rows, int slices, hipStream_t stream) {
  switch (H) {
    case 64: return h32_launch_sweep<2, 8, FWD>(pk, p, own_rows, slices, stream);
    case 128:
      if constexpr (FWD) return h32_launch_sweep<4, 8, true>(pk, p, own_rows, slices, stream);
      else return h32_launch_sweep<4, 4, false>(pk, p, own_rows, slices, stream);
    case 256: return h32_launch_sweep<8, 4, FWD>(pk, p, own_rows, slices, stream);
    default: b4r_set_error("head32: hidden size %d not supported", H); return B4R_E_SHAPE;
  }
}
}  // namespace

int b4r_head32_fwd_launch(const float* T, const float* E, const float* bias, int M, int V, int H, float* scratch, hipStream_t stream) {
  const int slices = b4r_head32_fwd_slices(M, V, H), tiles = b4r_cdiv(V, 32);
  char* recs = reinterpret_cast<char*>(scratch + up4l((int64_t)slices * M * (H + 8 + 2)));
  H32PackP pk{};
  pk.src = E; pk.R = V; pk.dst = recs; pk.mode = 0; pk.np = H / 32; pk.bias = bias;
  H32P p{};
  p.own = T; p.recs = recs; p.n_own = M; p.n_tiles = tiles; p.tiles_per_slice = b4r_cdiv(tiles, slices); p.part = scratch; p.M = M; p.V = V;
  const int rc = h32_dispatch<true>(H, pk, p, M, slices, stream);
  if (rc) return rc;
  B4R_CHECK_LAUNCH("masked-LM head forward (fused)");
  return B4R_OK;
}

// slabs of dE [slices][V][H] and of db [slices][V] into scratch (the caller reduces them); lse / ylab given, or (fwd_part != NULL)
// formed from the forward's compact (max, sum) pairs and the labels y
// the conversion of the transform rows for dE as a job another launch can carry (b4r_zero2's rider): *out is an H32PackP
int b4r_head32_dE_pack_job(const float* T, const float* lse, const int32_t* ylab, int M, int V, int H, float* scratch, const float* fwd_part,
                           int fwd_slices, const int64_t* y, void* out, size_t out_bytes, int* blocks) {
  if (out_bytes < sizeof(H32PackP)) return B4R_E_BADARG;
  const int slices = b4r_head32_dE_slices(M, V, H);
  H32PackP pk{};
  pk.src = T; pk.R = M; pk.dst = reinterpret_cast<char*>(scratch + up4l((int64_t)slices * ((int64_t)V * H + V))); pk.mode = 1; pk.np = H / 32;
  pk.lse = lse; pk.ylab = ylab; pk.V = V;
  if (fwd_part != nullptr) { pk.cpart = fwd_part + (int64_t)fwd_slices * M * (H + 8); pk.cslices = fwd_slices; pk.y = y; }
  *reinterpret_cast<H32PackP*>(out) = pk;
  *blocks = b4r_cdiv(M, 32);
  return B4R_OK;
}

int b4r_head32_dE_launch(const float* T, const float* E, const float* bias, const float* lse, const int32_t* ylab, int M, int V, int H,
                         float* scratch, hipStream_t stream, const float* fwd_part, int fwd_slices, const int64_t* y, int records_ready) {
  const int slices = b4r_head32_dE_slices(M, V, H), tiles = b4r_cdiv(M, 32);
  char* recs = reinterpret_cast<char*>(scratch + up4l((int64_t)slices * ((int64_t)V * H + V)));
  H32PackP pk{};
  pk.src = records_ready ? nullptr : T; pk.R = M; pk.dst = recs; pk.mode = 1; pk.np = H / 32; pk.lse = lse; pk.ylab = ylab; pk.V = V;
  if (fwd_part != nullptr) { pk.cpart = fwd_part + (int64_t)fwd_slices * M * (H + 8); pk.cslices = fwd_slices; pk.y = y; }
  H32P p{};
  p.own = E; p.recs = recs; p.n_own = V; p.n_tiles = tiles; p.tiles_per_slice = b4r_cdiv(tiles, slices); p.M = M; p.V = V; p.bias = bias;
  p.slab = scratch; p.bslab = scratch + (int64_t)slices * V * H;
  const int rc = h32_dispatch<false>(H, pk, p, V, slices, stream);
  if (rc) return rc;
  B4R_CHECK_LAUNCH("masked-LM head dE (fused)");
  return B4R_OK;
}
