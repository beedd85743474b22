// Attention half of a Keras TransformerEncoderBlock (post-LN), forward and backward, ONE launch each, one workgroup per sequence
// (hidden size 64, two heads of 32, L <= 224, split-precision bf16x3 arithmetic) -- round 3's rebuild of b4r_attn_block.hip on
// 32-token tiles:
//
//   * a wave owns 32 tokens (L = 200: 7 waves, two per SIMD, 256 registers each; round 2: 13 waves of 16 tokens under a 128-register
//     cap, 17-21 of them spilled) and works on 32 x 32 score blocks with v_mfma_f32_32x32x16_bf16: a quarter of the instructions per
//     score element, twice the matrix work per LDS fragment byte;
//   * the index a product sums over is always put on the accumulator ROWS of the product before it (b4r_tile32.h), so accumulators
//     feed the next product from registers; in the backward only dS crosses LDS (wave-private, 4 KB);
//   * backward, key-owner schedule: wave w keeps K_w, V_w (operand form) and the dK_w, dV_w accumulators in registers for a whole head
//     and walks the query tiles (w + s) mod NT; only the dQ partial of a step is added into the query tile's accumulator in LDS (one
//     conflict-free 16-byte read-modify-write per 4 registers, ordered by the step barrier => bitwise reproducible);
//   * x and dz1 rows are read once per head by the wave that owns them, straight into operand registers;
//   * attention-dropout decisions travel as one 32-bit word per (query, key tile): the backward loads them with scalar loads and
//     uses them as lane masks (v_cndmask with an SGPR pair): one instruction per element.
//
// Reference: tfm TransformerEncoderBlock / Keras MultiHeadAttention as constructed at
// bert4rec/models/components/networks/bert4rec_encoder.py:136-147 and called at :220-222 (key-padding mask -1e9,
// query scaled by 1/sqrt(d) after its bias, attention dropout on the probabilities, output dropout, residual,
// self_attention_layer_norm); the backward is tape.gradient of that call (bert4rec_model.py:166-167).
#include "b4r_tile32.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
// timing experiments only (tools/build_variant.sh ... -DA32_EXP=..): 1 no exp / dropout arithmetic, 2 no hi / lo split of Pd and dS,
// 4 no dS scratch + dQ product + accumulator update, 8 no barrier per step, 16 no dV / dK products, 32 no S / dA products
#ifndef A32_EXP
#define A32_EXP 0
#endif

// A32_PROF (tools/build_variant.sh ... -DA32_PROF): lane 0 of wave 0 of workgroup 0 stamps the shader clock at phase boundaries
#ifdef A32_PROF
__device__ long long g_a32_prof[64];
__device__ long long g_a32_sweep[2][64];
__device__ long long g_a32f_prof[32];
#define A32F_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_a32f_prof[k] = clock64(); } while (0)
#define A32_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_a32_prof[k] = clock64(); } while (0)
// stamps inside the sweep of head 0, by lane 0 of waves 0 and 4
#define A32_SWEEP(k) do { if (blockIdx.x == 0 && hd == 0 && (threadIdx.x & 255) == 0 && (k) < 64) g_a32_sweep[threadIdx.x >> 8][k] = clock64(); } while (0)
#else
#define A32_MARK(k) do { } while (0)
#define A32_SWEEP(k) do { } while (0)
#define A32F_MARK(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------------
// weights: 32 x 32 fp32 blocks -> hi / lo panel tiles (tile gt at img + gt * P_TILE).  src(gt) = address of the block's element
// (0, 0), ld(gt) its row pitch.  One float4 per thread and turn; the loads of TURNS turns are requested before the first conversion.
// ---------------------------------------------------------------------------------------------------------------------------
template <int TURNS, typename SrcFn>
__device__ __forceinline__ void stage_tiles(char* img, int ntiles, int nthreads, SrcFn src) {
  const int total = ntiles * 256;
  for (int f0 = threadIdx.x; f0 < total; f0 += TURNS * nthreads) {
    f32x4 v[TURNS];
    int dst[TURNS];
#pragma unroll
    for (int u = 0; u < TURNS; ++u) {
      const int f = min(f0 + u * nthreads, total - 1);   // clamped: every load unconditional
      const int gt = f >> 8, row = (f >> 3) & 31, q4 = f & 7;
      int ld;
      const float* s0 = src(gt, ld);
      v[u] = *reinterpret_cast<const f32x4*>(s0 + (int64_t)row * ld + 4 * q4);
      dst[u] = gt * P_TILE + p_chunk(row, q4 >> 1) + 8 * (q4 & 1);
    }
#pragma unroll
    for (int u = 0; u < TURNS; ++u) {
      if (f0 + u * nthreads < total) {
        bf16x4 hh, ll;
        b4r_split4(v[u], hh, ll);
        *reinterpret_cast<bf16x4*>(img + dst[u]) = hh;
        *reinterpret_cast<bf16x4*>(img + dst[u] + P_IMG) = ll;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// ROW layout of a wave's 32 x 64 fp32 block of a [N, 64] tensor: turn i (0..7) = rows 4i + (lane >> 4), columns 4 (lane & 15) .. +3
// (a wave-instruction moves 1 KB of consecutive addresses); rows beyond L are clamped to the last row.
// ---------------------------------------------------------------------------------------------------------------------------
// Addresses are a wave-uniform base (the sequence's first row) + a 32-bit lane offset: global_load saddr + voffset, no 64-bit lane
// arithmetic to keep in registers.
struct RowLay { int row, c4, wave32, L; };
__device__ __forceinline__ RowLay row_lay(int lane, int wave, int L) {
  RowLay k;
  k.row = lane >> 4; k.c4 = lane & 15; k.wave32 = 32 * wave; k.L = L;
  return k;
}
__device__ __forceinline__ uint32_t rl_off(const RowLay& k, int i) { return (uint32_t)(min(k.wave32 + 4 * i + k.row, k.L - 1) * HID + 4 * k.c4); }
__device__ __forceinline__ f32x4 ld4(const float* base, uint32_t off) { return *reinterpret_cast<const f32x4*>(base + off); }
__device__ __forceinline__ void st4(float* base, uint32_t off, const f32x4 v) { *reinterpret_cast<f32x4*>(base + off) = v; }
// ... -> bf16 hi / lo images of two column panels (tile0: columns 0..31, tile1: 32..63), natural column order
__device__ __forceinline__ void rl_to_panels(const RowLay& k, char* tile0, char* tile1, int i, const f32x4 v) {
  bf16x4 hh, ll;
  b4r_split4(v, hh, ll);
  char* d8 = ((k.c4 & 8) ? tile1 : tile0) + p_chunk(4 * i + k.row, (k.c4 & 7) >> 1) + 8 * (k.c4 & 1);
  *reinterpret_cast<bf16x4*>(d8) = hh;
  *reinterpret_cast<bf16x4*>(d8 + P_IMG) = ll;
}
// accumulator layout (rows 32 rt + (t & 3) + 8 (t >> 2) + 4h = columns of the tensor, lane r = token) -> row layout, through 8 KB of
// the wave's own LDS: two [32 tokens][32 columns] fp32 halves, 16-byte chunk c of row `row` at chunk position c ^ (row & 7)
// (conflict-free both ways)
// KORD: the accumulators' registers 8s .. 8s+7 are columns 16s + 8h + (0..7) (the output of a product whose A operand was a
// transposed read of an image in swapped column order) instead of the D layout's rows
template <bool KORD = false>
__device__ __forceinline__ void acc_to_rl(const RowLay& k, char* halfA, char* halfB, int r, int h, const f32x16 (&v)[2], f32x4 (&out)[8]) {
#pragma unroll
  for (int gp = 0; gp < 4; ++gp) {
    const int c = KORD ? 4 * (gp >> 1) + 2 * h + (gp & 1) : 2 * gp + h;   // 16-byte chunk (4 columns) of registers 4 gp .. 4 gp + 3
    *reinterpret_cast<f32x4*>(halfA + r * 128 + 16 * (c ^ (r & 7))) = (f32x4){v[0][4 * gp], v[0][4 * gp + 1], v[0][4 * gp + 2], v[0][4 * gp + 3]};
    *reinterpret_cast<f32x4*>(halfB + r * 128 + 16 * (c ^ (r & 7))) = (f32x4){v[1][4 * gp], v[1][4 * gp + 1], v[1][4 * gp + 2], v[1][4 * gp + 3]};
  }
  const char* src = (k.c4 & 8) ? halfB : halfA;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + k.row;
    out[i] = *reinterpret_cast<const f32x4*>(src + row * 128 + 16 * ((k.c4 & 7) ^ (row & 7)));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------
// accumulator (rows = the image's columns) -> row `row` of a tile for lane half h (acc_to_rows with the row chosen per lane)
__device__ __forceinline__ void acc_rows_at(char* tile, int row, int h, const f32x16& v) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 hi, lo;
    split8(regs8(v, s), hi, lo);
    char* d = tile + p_chunk(row, 2 * s + h);
    *reinterpret_cast<bf16x8*>(d) = hi;
    *reinterpret_cast<bf16x8*>(d + P_IMG) = lo;
  }
}
__device__ __forceinline__ int a32_slot_perm(int r) { return (r & 24) | ((r & 3) << 1) | ((r >> 2) & 1); }   // query r of a tile -> its decision word

struct A32BwdP {
  const float* x; const float* dz1; const float* ctx; const float* lse; const uint32_t* bits; const int64_t* mask;
  const float* Wqkv; const float* bqkv; const float* Wo;
  const float* zprev; const float* meanp; const float* rstdp; const float* gprev;     // the LayerNorm that produced x
  const int64_t* ids; const float* table; const float* pos; int V;                    // ... or the embedding stage (ids != NULL)
  float* dqkv; float* da; float* ln_part;
  float* dw_slab; float* db_slab;   // per-sequence partials of dWqkv [B][64][192] and dbqkv [B][192] (or NULL: dqkv is written instead)
  float* dwo_slab; float* dbo_slab; // ... of dWo [B][64][64] and dbo [B][64] (or NULL)
  // the rows of dz1 that carry a gradient (or NULL: all of them): row clamp(slot_pos[b][j]) where slot_ids[b][j] != 0, j < slots; every
  // other row of dz1 counts as zero and is never read (the last layer of a train step: only the masked-LM rows)
  const int64_t* slot_pos; const int64_t* slot_ids; int slots;
  int B, L, NT;
  float qscale;
  DropArgs drop_p, drop_o, drop_e;
};

// LDS of the backward (bytes): [Q images NT x 4 KB | dO images | dQ accumulators (fp32, register layout) | per-wave scratch (K^T
// staging, then dS) | weight slices of one head 32 KB | key mask adders, -lse, D, biases, LayerNorm partials]
__host__ __device__ constexpr int bwd32_small_floats(int NT) { return 3 * NT * 32 + 192 + NT * 128 + 8 + NT * 96 + NT * 32; }
__host__ __device__ constexpr int bwd32_lds(int NT) { return 4 * NT * P_TILE + 8 * P_TILE + bwd32_small_floats(NT) * 4; }
// CQ (compact queries, below): + token -> slot [NT x 32], slot -> token [64], the sweep's decision words [NT key tiles][2][32]
__host__ __device__ constexpr int bwd32_lds_cq(int NT) { return bwd32_lds(NT) + (NT * 32 + 64 + NT * 64) * 4; }

// CQ (the last layer of a train step, p.slot_pos != NULL, at most 64 slots): only the labelled slots' rows of dz1 carry a gradient,
// so only those QUERIES have a non-zero dO -- every other query's softmax backward is exactly zero.  The sweep then walks one or two
// COMPACT query tiles (query j = slot j; rows of padded slots are empty: -lse = -inf) instead of the NT token tiles: every wave still
// projects q~ / dO of its own tokens, the labelled ones scatter their rows into the compact Q~ / dO images, a key owner walks the compact
// tiles itself (no rotating ownership), keeps its dQ partials in registers, the partials are summed over the key owners in wave order
// and each token picks up its slot's dq row (zero without a slot).  The forward's decision words are regathered into the compact order.
template <bool EMBED, bool DROP, bool CQ = false>
__global__ __launch_bounds__(512, 2) void attn32_bwd_kernel(A32BwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int NT = p.NT, L = p.L;
  char* const QIMG = smem32;
  char* const DOIMG = QIMG + NT * P_TILE;
  char* const DQACC = DOIMG + NT * P_TILE;
  char* const SCRALL = DQACC + NT * P_TILE;
  char* const WIMG = SCRALL + NT * P_TILE;            // [Wq 2 tiles | Wk 2 | Wv 2 | Wo rows of the head: 2 panels]
  float* const sAdd = reinterpret_cast<float*>(WIMG + 8 * P_TILE);   // (mask adder - amax) * log2e per key, -inf beyond L
  float* const sCS = sAdd + NT * 32;                  // -lse * log2e per query of this head (-inf: pad)
  float* const sD = sCS + NT * 32;                    // rowsum(dctx * ctx) per query of this head
  float* const sbq = sD + NT * 32;                    // bqkv [192]
  float* const sred = sbq + 192;                      // [NT][128] LayerNorm partials
  int* const sflag = reinterpret_cast<int*>(sred + NT * 128);   // [NT] steps of the sweep each wave has finished
  float* const sdb = reinterpret_cast<float*>(sflag + 8);      // [NT][96] column sums of dq | dk | dv over a wave's tokens
  float* const sHas = sdb + NT * 96;                            // [NT * 32] 1 where the token's dz1 row carries a gradient
  int* const sSlot = reinterpret_cast<int*>(sHas + NT * 32);   // CQ: [NT * 32] the token's slot, -1: none
  int* const sTok = sSlot + NT * 32;                            // CQ: [64] the slot's token, -1: padded slot
  uint32_t* const sCW = reinterpret_cast<uint32_t*>(sTok + 64); // CQ: [NT][2][32] decision words of (key tile, compact query tile)
  const int NQ = CQ ? (p.slots + 31) >> 5 : 0;

  const int nthreads = blockDim.x;
  const int b = blockIdx.x;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const scr = SCRALL + wave * P_TILE;
  // Lane constants are derived from an OPAQUE copy of the lane id inside each head pass and again in the epilogue: hipcc otherwise
  // hoists every lane-invariant address, dropout hash and fragment offset of both heads to the kernel's entry and spills them (70-100
  // registers in scratch); recomputing them costs ~60 vector instructions per pass.
#define A32_LANE_CONSTS()                                                           \
  int ln = lane;                                                                    \
  asm volatile("" : "+v"(ln));                                                      \
  const Lane32 lk = lane32(ln);                                                     \
  const int r = lk.r, h = lk.h;                                                     \
  const int tok = 32 * wave + r, tokc = min(tok, L - 1);                            \
  const bool live = tok < L;                                                        \
  const RowLay rl = row_lay(ln, wave, L);                                           \
  const int rl_row = rl.row, rl_c4 = rl.c4;                                         \
  (void)tokc; (void)live; (void)rl_row; (void)rl_c4
  // turn i of the wave's dz1 rows (row layout); rows without a gradient are zero and not read
#define A32_DZ_ROW()                                                                \
  auto dz_row = [&](int i_) __attribute__((always_inline)) -> f32x4 {              \
    const bool has_ = sHas[min(32 * wave + 4 * i_ + rl.row, L - 1)] != 0.f;         \
    return has_ ? ld4(dzb, rl_off(rl, i_)) : (f32x4){0.f, 0.f, 0.f, 0.f};          \
  };                                                                                \
  (void)dz_row

  // key mask: one turn (at least 32 threads per 32 tokens)
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  for (int k = threadIdx.x; k < 192; k += nthreads) sbq[k] = p.bqkv[k];
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
  const bool dead = amax != 0.0f;   // every key masked: Keras' -1e9 absorbs the scores, the softmax is uniform over all L keys
  if ((int)threadIdx.x < NT * 32) {
    sAdd[threadIdx.x] = (int)threadIdx.x < L ? (((1.0f - (float)mval) * -1e9f) - amax) * LOG2E : -INFINITY;
    sHas[threadIdx.x] = p.slot_pos != nullptr ? 0.f : 1.f;
    if (CQ) { sSlot[threadIdx.x] = -1; if (threadIdx.x < 64) sTok[threadIdx.x] = -1; }
  }
  if (p.slot_pos != nullptr) {   // (block-uniform)
    lds_barrier();
    for (int j = threadIdx.x; j < p.slots; j += nthreads)
      if (p.slot_ids[(int64_t)b * p.slots + j] != 0) {
        const int64_t q = p.slot_pos[(int64_t)b * p.slots + j];
        const int qc = q < 0 ? 0 : (q >= L ? L - 1 : (int)q);
        sHas[qc] = 1.f;
        if (CQ) { sSlot[qc] = j; sTok[j] = qc; }
      }
  }
  lds_barrier();

  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  const DropCtx dco = b4r_drop_ctx(p.drop_o);
  const float pscale = DROP ? dcp.scale : 1.0f;

  // the wave's 32 x 64 fp32 block of a [N, 64] tensor in ROW layout (RowLay): 1 KB contiguous per wave-instruction
  const float* const xb = p.x + row0 * HID;       // wave-uniform bases of the sequence's rows
  const float* const dzb = p.dz1 + row0 * HID;
  const float* const ctxb = p.ctx + row0 * HID;
  float* const dab = p.da + row0 * HID;
  float* const dqkvb = p.dqkv + row0 * (3 * HID);
  char* const ownA = QIMG + wave * P_TILE;
  char* const ownB = DOIMG + wave * P_TILE;
  char* const ownC = DQACC + wave * P_TILE;

  // ---- dWo = ctx^T . dropmask(dz1) and dbo, before the heads: ctx and dropmask(dz1) rows of the whole sequence as images (two
  // column panels each, in the waves' own tiles), output tile (context half, hidden half) = wave 0..3 over all token tiles.  The
  // dropmask(dz1) images stay where the first head's projections expect them.
  const bool fold_wo = p.dwo_slab != nullptr;
  if (fold_wo) {
    A32_LANE_CONSTS();
    A32_DZ_ROW();
    f32x4 dbo = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool lv = 32 * wave + 4 * i + rl_row < L;
      f32x4 dy = dz_row(i);
      // (CQ: the forward wrote the slots' rows of ctx only -- the others are neither defined nor needed: their dy is zero)
      const f32x4 cr = (CQ && sHas[min(32 * wave + 4 * i + rl_row, L - 1)] == 0.f) ? (f32x4){0.f, 0.f, 0.f, 0.f} : ld4(ctxb, rl_off(rl, i));
      if (dco.on) dy = b4r_drop4(dco, dy, (uint64_t)(row0 + 32 * wave + 4 * i + rl_row) * HID + (uint64_t)(4 * rl_c4));
      if (!lv) dy = (f32x4){0.f, 0.f, 0.f, 0.f};   // pad tokens: no contribution
      dbo += dy;
      rl_to_panels(rl, ownC, scr, i, dy);
      rl_to_panels(rl, ownA, ownB, i, cr);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { dbo[e] += other_row(dbo[e]); dbo[e] += other_half(dbo[e], h); }
    if (lane < 16) *reinterpret_cast<f32x4*>(&sdb[wave * 96 + 4 * rl_c4]) = dbo;
    lds_barrier();
    for (int tile = wave; tile < 4; tile += NT) {
      const int ci = tile >> 1, hj = tile & 1;
      const char* cimg = ci ? DOIMG : QIMG;
      const char* yimg = hj ? SCRALL : DQACC;
      f32x16 acc = zero16();
      for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const char* ca = cimg + kt * P_TILE;
          const char* yb = yimg + kt * P_TILE;
          acc = mfma32x3(tr_pair(ca + lk.trn[s][0], ca + lk.trn[s][1]), tr_pair(ca + P_IMG + lk.trn[s][0], ca + P_IMG + lk.trn[s][1]),
                         tr_pair(yb + lk.trn[s][0], yb + lk.trn[s][1]), tr_pair(yb + P_IMG + lk.trn[s][0], yb + P_IMG + lk.trn[s][1]), acc);
        }
      }
      float* dst = p.dwo_slab + (int64_t)b * (HID * HID) + 32 * hj + r;
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) dst[(32 * ci + (tt & 3) + 8 * (tt >> 2) + 4 * h) * HID] = acc[tt];
    }
    for (int k = threadIdx.x; k < HID; k += nthreads) {
      float acc = 0.f;
      for (int w = 0; w < NT; ++w) acc += sdb[w * 96 + k];
      p.dbo_slab[(int64_t)b * HID + k] = acc;
    }
  }

  f32x16 dx[2];   // dX^T[hidden 32 rt + ..][token] of ONE head: the first head's waits in `da` (the wave's own rows) for the epilogue --
                  // 32 registers that the sweep of the second head needs

  // x and dz1 rows of the wave's tokens (row layout, coalesced), lse and the ctx columns of a head are REQUESTED one head ahead: the
  // second head's while the first one's weight-gradient section still has its slab stores to issue (a load behind those stores waits for
  // their acknowledgements: 11 k cycles at the top of the second head)
  f32x4 nxr[8], nyr[8], ncx[4];
  float nlse;
#define A32_REQUEST(HD_, SKIP_DY_)                                                                                      \
  do {                                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { nxr[i] = ld4(xb, rl_off(rl, i)); nyr[i] = (SKIP_DY_) ? nxr[i] : dz_row(i); } \
    nlse = live ? (p.lse + ((int64_t)b * 2 + (HD_)) * L)[(uint32_t)tok] : INFINITY;                                    \
    _Pragma("unroll") for (int gp = 0; gp < 4; ++gp)                                                                    \
      ncx[gp] = (CQ && sHas[tokc] == 0.f) ? (f32x4){0.f, 0.f, 0.f, 0.f} : ld4(ctxb, (uint32_t)(tokc * HID + 32 * (HD_) + 4 * h + 8 * gp)); \
  } while (0)
  {
    A32_LANE_CONSTS();
    A32_DZ_ROW();
    A32_REQUEST(0, fold_wo);   // (the dWo section left the dropmask(dz1) images in place)
  }
  A32_MARK(0);
  for (int hd = 0; hd < 2; ++hd) {
    A32_LANE_CONSTS();
    A32_DZ_ROW();
    const int64_t bh = (int64_t)b * 2 + hd;
    f32x4 xr[8], yr[8], cx[4];
    const bool dy_staged = fold_wo && hd == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { xr[i] = nxr[i]; yr[i] = nyr[i]; }
#pragma unroll
    for (int gp = 0; gp < 4; ++gp) cx[gp] = ncx[gp];
    const float lse_q = nlse;
    {   // the carried requests are consumed: nothing of them stays live through the sweep (the compiler cannot know that the loop ends
        // after the pass that does not refill them)
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) { nxr[i] = z4; nyr[i] = z4; }
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) ncx[gp] = z4;
      nlse = 0.f;
    }
    A32_MARK(1 + 10 * hd);
    lds_barrier();   // every wave is done with the previous head's weight slices (the wave's own tiles were free before)
    if (CQ && threadIdx.x < 64) { sCS[threadIdx.x] = -INFINITY; sD[threadIdx.x] = 0.f; }   // (padded slots: empty query rows)
    A32_MARK(2 + 10 * hd);
    // tiles 0..5: W_j[hidden 32 rt ..][features 32 hd ..] (j = q, k, v; tile 2j + rt); 6, 7: Wo[32 hd ..][hidden 32 (gt - 6) ..]
    stage_tiles<5>(WIMG, 8, nthreads, [&](int gt, int& ld) __attribute__((always_inline)) -> const float* {
      const bool wo = gt >= 6;
      ld = wo ? HID : 3 * HID;
      return wo ? p.Wo + (int64_t)32 * hd * HID + 32 * (gt - 6) : p.Wqkv + (int64_t)32 * (gt & 1) * (3 * HID) + HID * (gt >> 1) + 32 * hd;
    });
    // x -> images in the wave's Q / dO tiles, dropmask(dz1) -> images in its dQ-accumulator tile and scratch
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      rl_to_panels(rl, ownA, ownB, i, xr[i]);
      if (dy_staged) continue;
      f32x4 dy = yr[i];
      if (dco.on) dy = b4r_drop4(dco, dy, (uint64_t)(row0 + 32 * wave + 4 * i + rl_row) * HID + (uint64_t)(4 * rl_c4));
      rl_to_panels(rl, ownC, scr, i, dy);
    }
    A32_MARK(3 + 10 * hd);
    lds_barrier();
    A32_MARK(4 + 10 * hd);

    // ---- q~, k, v of this head (recomputed) and dctx = dropmask(dz1).Wo^T, transposed: rows = feature, lane = token -------------
    f32x16 qT, kT, vT, dcT;
    {
      const f32x16 bq = rows_of(sbq + 32 * hd, h), bk = rows_of(sbq + HID + 32 * hd, h), bv = rows_of(sbq + 2 * HID + 32 * hd, h);
      qT = bq; kT = bk; vT = bv; dcT = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const char* xt = ((ks >> 1) ? ownB : ownA) + lk.rowc[ks & 1];    // B = x^T[k = hidden][token]: chunk 2 (ks & 1) + h of row `token`
        const char* yt = ((ks >> 1) ? scr : ownC) + lk.rowc[ks & 1];
        const bf16x8 xh = row_at(xt), xl = row_at(xt + P_IMG), yh = row_at(yt), yl = row_at(yt + P_IMG);
        const char* wq = WIMG + (ks >> 1) * P_TILE;
        const int a0 = lk.trn[ks & 1][0], a1 = lk.trn[ks & 1][1];
        qT = mfma32x3(tr_pair(wq + a0, wq + a1), tr_pair(wq + P_IMG + a0, wq + P_IMG + a1), xh, xl, qT);
        const char* wk = wq + 2 * P_TILE;
        kT = mfma32x3(tr_pair(wk + a0, wk + a1), tr_pair(wk + P_IMG + a0, wk + P_IMG + a1), xh, xl, kT);
        const char* wv = wq + 4 * P_TILE;
        vT = mfma32x3(tr_pair(wv + a0, wv + a1), tr_pair(wv + P_IMG + a0, wv + P_IMG + a1), xh, xl, vT);
        const char* wo = WIMG + (6 + (ks >> 1)) * P_TILE + lk.rowc[ks & 1];   // A = Wo[row = context column][k = hidden], by rows
        dcT = mfma32x3(row_at(wo), row_at(wo + P_IMG), yh, yl, dcT);
      }
      qT = qT * (p.qscale * LOG2E);
    }
    A32_MARK(5 + 10 * hd);
    // D = sum_c dctx * ctx over this head's 32 columns (the softmax backward's row term)
    {
      float d = 0.f;
#pragma unroll
      for (int gp = 0; gp < 4; ++gp)
#pragma unroll
        for (int e = 0; e < 4; ++e) d = fmaf(dcT[4 * gp + e], cx[gp][e], d);
      d += other_half(d, h);
      if (!CQ) { if (h == 0) { sD[tok] = d; sCS[tok] = -lse_q * LOG2E; } }
      else { const int js_ = live ? sSlot[tok] : -1; if (h == 0 && js_ >= 0) { sD[js_] = d; sCS[js_] = -lse_q * LOG2E; } }
    }
    // operand forms and images of the wave's tile (the x / dz1 images are dead: every read of them is in front of these writes)
    bf16x8 kBh[2], kBl[2], vBh[2], vBl[2];     // K^T, V^T [feature][key] as B operands of S = Q~.K^T and dA = dO.V^T
#pragma unroll
    for (int s = 0; s < 2; ++s) { acc_frag(kT, s, kBh[s], kBl[s]); acc_frag(vT, s, vBh[s], vBl[s]); }
    const int js = (CQ && live) ? sSlot[tok] : -1;
    if (!CQ) {
      acc_to_rows(ownA, lk, qT);
      acc_to_rows(ownB, lk, dcT);
    } else {
      lds_barrier();   // the compact rows go into OTHER waves' tiles: every wave has finished the projections' reads of its x / dz1 images
      if (js >= 0) {
        acc_rows_at(QIMG + (js >> 5) * P_TILE, js & 31, h, qT);
        acc_rows_at(DOIMG + (js >> 5) * P_TILE, js & 31, h, dcT);
      }
      // the decision words of (key tile w, compact tile): slot j's word is its token's, in the slot's place
      for (int i = threadIdx.x; i < NT * 64; i += nthreads) {
        const int w_ = i >> 6, jj = i & 63, tq = sTok[jj];
        uint32_t word = 0;
        if (DROP && tq >= 0) word = p.bits[((bh * NT + w_) * NT + (tq >> 5)) * 32 + a32_slot_perm(tq & 31)];
        sCW[(w_ * 2 + (jj >> 5)) * 32 + a32_slot_perm(jj & 31)] = word;
      }
    }
    acc_to_rows(scr, lk, kT);
    bf16x8 kTh[2], kTl[2];                     // K^T[feature position][key] as the A operand of dQ^T = K^T.dS^T
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kTh[s] = tr_pair(scr + lk.trn[s][0], scr + lk.trn[s][1]);
      kTl[s] = tr_pair(scr + P_IMG + lk.trn[s][0], scr + P_IMG + lk.trn[s][1]);
    }
    if (!CQ) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(ownC + j * 1024 + lane * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    A32_MARK(6 + 10 * hd);
    if (lane == 0) sflag[wave] = 0;
    lds_barrier();   // images, D, -lse of every tile in place; accumulators zero; (the K^T reads above have landed: lgkmcnt(0))
    A32_MARK(7 + 10 * hd);

    // ---- the sweep: wave w (key tile w) x query tile (w + s) mod NT.  A step is a VECTOR phase (S, dA of this tile -> Pd, dS as
    // operand fragments; dS also to the wave's scratch image) and a MATRIX phase of ten 3-instruction products issued back to back:
    // dV, dK (two query halves each), S, dA of the NEXT tile (two feature halves each), the dQ partial (two key halves).  Every
    // product's LDS fragments are requested two products ahead (hipcc otherwise puts each read next to its use: a wave then waits
    // out the LDS latency in front of every product, and two waves per SIMD cannot cover that).  No barrier inside the sweep: the dQ
    // accumulator of tile t was last touched by wave w + 1 in ITS previous step, a flag per wave orders that (fixed order of the
    // additions => bitwise reproducible).
    f32x16 dK = zero16(), dV = zero16(), S, dA;
    const float adk = sAdd[tok];
    if (dead) {   // S = -lse for every key: the scores are absorbed by the mask adder
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { kBh[ks][e] = (__bf16)0.f; kBl[ks][e] = (__bf16)0.f; }
    }
#define A32_LD_TR(img, s2, H, Lo)                                                \
  H = tr_pair((img) + lk.trp[s2][0], (img) + lk.trp[s2][1]);                     \
  Lo = tr_pair((img) + P_IMG + lk.trp[s2][0], (img) + P_IMG + lk.trp[s2][1])
#define A32_LD_ROW(img, ks, H, Lo)                                               \
  H = row_at((img) + lk.rowc[ks]);                                               \
  Lo = row_at((img) + P_IMG + lk.rowc[ks])
#define A32_LD_SCR(ks, H, Lo)                                                    \
  H = tr_pair(scr + lk.trn[ks][0], scr + lk.trn[ks][1]);                         \
  Lo = tr_pair(scr + P_IMG + lk.trn[ks][0], scr + P_IMG + lk.trn[ks][1])
#define A32_SB() __builtin_amdgcn_sched_barrier(0)
    auto tile_of = [&](int s) __attribute__((always_inline)) { const int t = wave + s; return t >= NT ? t - NT : t; };
    typedef const __attribute__((address_space(4))) uint64_t* kmask_ptr;   // constant address space: scalar loads
    if (CQ) {
      f32x16 dQp[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dQp[t] = zero16();
        if (t >= NQ) continue;   // (block-uniform)
        const char* qimg = QIMG + t * P_TILE;
        const char* dimg = DOIMG + t * P_TILE;
        S = rows_of(sCS + 32 * t, h);
        dA = zero16();
#pragma unroll
        for (int e = 0; e < 16; ++e) S[e] += adk;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 ah, al;
          A32_LD_ROW(qimg, ks, ah, al);
          S = mfma32x3(ah, al, kBh[ks], kBl[ks], S);
          A32_LD_ROW(dimg, ks, ah, al);
          dA = mfma32x3(ah, al, vBh[ks], vBl[ks], dA);
        }
        const f32x16 Dq = rows_of(sD + 32 * t, h);
        uint64_t km[16];
        if (DROP) {
          const uint32_t* cw = sCW + (wave * 2 + t) * 32;
#pragma unroll
          for (int tt = 0; tt < 16; ++tt) {
            const uint32_t lo_ = __builtin_amdgcn_readfirstlane(cw[2 * tt]), hi_ = __builtin_amdgcn_readfirstlane(cw[2 * tt + 1]);
            km[tt] = ((uint64_t)hi_ << 32) | lo_;
          }
        }
        bf16x8 pdh[2], pdl[2], dsh[2], dsl[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          f32x8 pd, ds;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int tt = 8 * s2 + j;
            const float pr = __builtin_amdgcn_exp2f(S[tt]);
            if (DROP) {
              const float kf = __builtin_amdgcn_inverse_ballot_w64(km[tt]) ? pscale : 0.f;
              pd[j] = pr * kf;
              ds[j] = pr * fmaf(dA[tt], kf, -Dq[tt]);
            } else {
              pd[j] = pr;
              ds[j] = pr * (dA[tt] - Dq[tt]);
            }
          }
          split8(pd, pdh[s2], pdl[s2]);
          split8(ds, dsh[s2], dsl[s2]);
          const s16x8 hv = __builtin_bit_cast(s16x8, dsh[s2]), lv = __builtin_bit_cast(s16x8, dsl[s2]);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            char* w8 = scr + p_chunk(r, 2 * s2 + a) + 8 * h;
            *reinterpret_cast<s16x4*>(w8) = a ? __builtin_shufflevector(hv, hv, 4, 5, 6, 7) : __builtin_shufflevector(hv, hv, 0, 1, 2, 3);
            *reinterpret_cast<s16x4*>(w8 + P_IMG) = a ? __builtin_shufflevector(lv, lv, 4, 5, 6, 7) : __builtin_shufflevector(lv, lv, 0, 1, 2, 3);
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 fh, fl;
          A32_LD_TR(dimg, s2, fh, fl);
          dV = mfma32x3(fh, fl, pdh[s2], pdl[s2], dV);
          A32_LD_TR(qimg, s2, fh, fl);
          dK = mfma32x3(fh, fl, dsh[s2], dsl[s2], dK);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 fh, fl;
          A32_LD_SCR(ks, fh, fl);
          dQp[t] = mfma32x3(kTh[ks], kTl[ks], fh, fl, dQp[t]);
        }
      }
      // dQ of compact tile t = the key owners' partials in wave order, left in accumulator tile t (register layout)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t >= NQ) continue;
        lds_barrier();   // (t == 0: every sweep is over; t > 0: the previous tile's partials have been read)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f32x4*>(scr + j * 1024 + lane * 16) = (f32x4){dQp[t][4 * j], dQp[t][4 * j + 1], dQp[t][4 * j + 2], dQp[t][4 * j + 3]};
        lds_barrier();
        if (wave == t) {
          f32x16 gsum = zero16();
          for (int w_ = 0; w_ < NT; ++w_)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 a4 = *reinterpret_cast<const f32x4*>(SCRALL + w_ * P_TILE + j * 1024 + lane * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) gsum[4 * j + e] += a4[e];
            }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(DQACC + t * P_TILE + j * 1024 + lane * 16) = (f32x4){gsum[4 * j], gsum[4 * j + 1], gsum[4 * j + 2], gsum[4 * j + 3]};
        }
      }
    } else {
    {   // S, dA of the first tile
      const char* qimg = QIMG + wave * P_TILE;
      const char* dimg = DOIMG + wave * P_TILE;
      S = rows_of(sCS + 32 * wave, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) S[e] += adk;
      dA = zero16();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah, al;
        A32_LD_ROW(qimg, ks, ah, al);
        S = mfma32x3(ah, al, kBh[ks], kBl[ks], S);
        A32_LD_ROW(dimg, ks, ah, al);
        dA = mfma32x3(ah, al, vBh[ks], vBl[ks], dA);
      }
    }
    for (int s = 0; s < NT; ++s) {
      const int t = tile_of(s), tn = tile_of(s + 1);   // (the last step forms S / dA of a tile nobody uses: no branch in the body)
      const char* qimg = QIMG + t * P_TILE;
      const char* dimg = DOIMG + t * P_TILE;
      const char* qn = QIMG + tn * P_TILE;
      const char* dn = DOIMG + tn * P_TILE;
      A32_SWEEP(2 + 4 * s);
      // ---- vector phase ------------------------------------------------------------------------------------------------------
      bf16x8 f0h, f0l, f1h, f1l;
      A32_LD_TR(dimg, 0, f0h, f0l);   // the first two products' fragments travel during the vector phase
      A32_LD_TR(qimg, 0, f1h, f1l);
      const f32x16 Dq = rows_of(sD + 32 * t, h);
      uint64_t km[16];   // keep decisions of the 32 x 32 block: register tt's lane mask over the keys
      if (DROP) {
        kmask_ptr mp = (kmask_ptr)(reinterpret_cast<const uint64_t*>(p.bits) + ((bh * NT + wave) * NT + t) * 16);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) km[tt] = mp[tt];
      }
      bf16x8 pdh[2], pdl[2], dsh[2], dsl[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f32x8 pd, ds;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int tt = 8 * s2 + j;
          const float pr = (A32_EXP & 1) ? S[tt] : __builtin_amdgcn_exp2f(S[tt]);
          if (A32_EXP & 1) { pd[j] = pr; ds[j] = dA[tt]; } else
          if (DROP) {
            const float kf = __builtin_amdgcn_inverse_ballot_w64(km[tt]) ? pscale : 0.f;
            pd[j] = pr * kf;
            ds[j] = pr * fmaf(dA[tt], kf, -Dq[tt]);
          } else {
            pd[j] = pr;
            ds[j] = pr * (dA[tt] - Dq[tt]);
          }
        }
        split8(pd, pdh[s2], pdl[s2]);
        split8(ds, dsh[s2], dsl[s2]);
        // dS -> the wave's [key][query] scratch image: registers 4a .. 4a+3 of this half are queries 16 s2 + 8a + 4h + (0..3)
        const s16x8 hv = __builtin_bit_cast(s16x8, dsh[s2]), lv = __builtin_bit_cast(s16x8, dsl[s2]);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          char* w8 = scr + p_chunk(r, 2 * s2 + a) + 8 * h;
          *reinterpret_cast<s16x4*>(w8) = a ? __builtin_shufflevector(hv, hv, 4, 5, 6, 7) : __builtin_shufflevector(hv, hv, 0, 1, 2, 3);
          *reinterpret_cast<s16x4*>(w8 + P_IMG) = a ? __builtin_shufflevector(lv, lv, 4, 5, 6, 7) : __builtin_shufflevector(lv, lv, 0, 1, 2, 3);
        }
      }
      // S of the next tile starts from -lse (per query) + the key's mask adder
      S = rows_of(sCS + 32 * tn, h);
#pragma unroll
      for (int e = 0; e < 16; ++e) S[e] += adk;
      A32_SWEEP(3 + 4 * s);
      A32_SB();
      // ---- matrix phase: ten products, fragments requested two products ahead ---------------------------------------------------
      bf16x8 f2h, f2l, f3h, f3l, f4h, f4l, f5h, f5l, f6h, f6l, f7h, f7l, f8h, f8l, f9h, f9l;
      A32_LD_TR(dimg, 1, f2h, f2l);
      dV = mfma32x3(f0h, f0l, pdh[0], pdl[0], dV);   // dV^T[feature][key] += dO^T[feature][query] . Pd[query][key], queries 0..15
      A32_SB();
      A32_LD_TR(qimg, 1, f3h, f3l);
      dK = mfma32x3(f1h, f1l, dsh[0], dsl[0], dK);   // dK^T += Q~^T . dS
      A32_SB();
      A32_LD_ROW(qn, 0, f4h, f4l);
      dV = mfma32x3(f2h, f2l, pdh[1], pdl[1], dV);
      A32_SB();
      A32_LD_ROW(dn, 0, f5h, f5l);
      dK = mfma32x3(f3h, f3l, dsh[1], dsl[1], dK);
      A32_SB();
      A32_LD_ROW(qn, 1, f6h, f6l);
      S = mfma32x3(f4h, f4l, kBh[0], kBl[0], S);     // S[query][key] = Q~ . K^T of the next tile
      A32_SB();
      A32_LD_ROW(dn, 1, f7h, f7l);
      dA = mfma32x3(f5h, f5l, vBh[0], vBl[0], zero16());
      A32_SB();
      A32_LD_SCR(0, f8h, f8l);
      S = mfma32x3(f6h, f6l, kBh[1], kBl[1], S);
      A32_SB();
      A32_LD_SCR(1, f9h, f9l);
      dA = mfma32x3(f7h, f7l, vBh[1], vBl[1], dA);
      A32_SB();
      // dQ^T[feature position][query] = K^T[.][key] . dS^T[key][query]  (B by transposed reads of the scratch image)
      f32x16 dQp = mfma32x3(kTh[0], kTl[0], f8h, f8l, zero16());
      dQp = mfma32x3(kTh[1], kTl[1], f9h, f9l, dQp);
      A32_SWEEP(4 + 4 * s);
      if (s > 0) {   // the accumulator's previous addition (wave w + 1, its step s - 1) must be in place
        const volatile int* f = sflag + (wave + 1 < NT ? wave + 1 : 0);
        while (*f < s) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      char* acc = DQACC + t * P_TILE + lane * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4* a4 = reinterpret_cast<f32x4*>(acc + j * 1024);
        *a4 = *a4 + (f32x4){dQp[4 * j], dQp[4 * j + 1], dQp[4 * j + 2], dQp[4 * j + 3]};
      }
      asm volatile("" ::: "memory");
      *reinterpret_cast<volatile int*>(sflag + wave) = s + 1;   // LDS operations of a wave complete in order: the sums are in place
      A32_SWEEP(5 + 4 * s);
    }
    }
    lds_barrier();   // every accumulator is complete

    // ---- results of this head for the wave's tokens: registers 8s .. 8s+7 = features 16s + 8h + (0..7) ----------------------------
    A32_MARK(8 + 10 * hd);
    f32x16 gq;
    if (!CQ) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ownC + j * 1024 + lane * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[4 * j + e] = a4[e] * p.qscale;
      }
    } else {   // the token's slot row of the compact dQ (lane (slot & 31, h) of tile slot >> 5), zero without a slot
      const char* src = DQACC + (max(js, 0) >> 5) * P_TILE + (32 * h + (max(js, 0) & 31)) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(src + j * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[4 * j + e] = js >= 0 ? a4[e] * p.qscale : 0.f;
      }
    }
    const f32x16 gk = dK * LN2;   // Q~ carries log2(e)
    const bool fold = p.dw_slab != nullptr;   // dWqkv / dbqkv formed here instead of writing dqkv for a weight-gradient launch
    if (live && p.dqkv != nullptr) {
      const uint32_t dst = (uint32_t)(tok * (3 * HID) + 32 * hd + 8 * h);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int o = 8 * s + 4 * hf;
          st4(dqkvb, dst + 16 * s + 4 * hf, (f32x4){gq[o], gq[o + 1], gq[o + 2], gq[o + 3]});
          st4(dqkvb, dst + HID + 16 * s + 4 * hf, (f32x4){gk[o], gk[o + 1], gk[o + 2], gk[o + 3]});
          st4(dqkvb, dst + 2 * HID + 16 * s + 4 * hf, (f32x4){dV[o], dV[o + 1], dV[o + 2], dV[o + 3]});
        }
      }
    }
    // dX^T[hidden][token] += W_j[hidden][feature] . g_j^T[feature][token], j = q, k, v: A by rows of the weight slices
    dx[0] = zero16(); dx[1] = zero16();
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x16& gsrc = j == 0 ? gq : (j == 1 ? gk : dV);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 gh, gl;
        acc_frag(gsrc, s, gh, gl);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const char* wj = WIMG + (2 * j + rt) * P_TILE + lk.rowc[s];
          dx[rt] = mfma32x3(row_at(wj), row_at(wj + P_IMG), gh, gl, dx[rt]);
        }
      }
    }
    A32_MARK(9 + 10 * hd);
    if (hd == 0) {   // the first head's dX waits in `da`, in row layout (the sweep is over: the wave's Q / dO tiles are free)
      f32x4 o[8];
      acc_to_rl(rl, ownA, ownB, r, h, dx, o);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (32 * wave + 4 * i + rl_row < L) st4(dab, rl_off(rl, i), o[i]);
    }
    if (fold) {
      A32_LANE_CONSTS();   // (shadows the pass's constants: nothing lane-derived has to survive the sweep for this section)
      // dWqkv[hidden][feature] = sum over the sequence's tokens of x^T . dqkv: the wave's dq, dk, dv rows go to images (its Q, dO and
      // scratch tiles; registers 8s .. 8s+7 are features 16s + 8h + .. : natural column order), x comes back as images of two column
      // panels (the wave's dQ-accumulator tile and its tile of the weight region), then wave w forms output tile w of the six
      // (j = q, k, v; hidden rows 32 rt ..) over ALL token tiles.  Pad tokens contribute zeros (their dq, dk, dv are zero).
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const f32x16& gsrc = j == 0 ? gq : (j == 1 ? gk : dV);
        acc_to_rows(j == 0 ? ownA : (j == 1 ? ownB : scr), lk, gsrc);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
          const float cs = half_sum31(gsrc[tt]);
          if (r == 31) sdb[wave * 96 + 32 * j + 16 * (tt >> 3) + 8 * h + (tt & 7)] = cs;
        }
      }
      f32x4 xr2[8];   // x rows again: the next head's request carries them (or, behind the second head, a request of their own)
      if (hd == 0) {
        A32_REQUEST(1, false);
#pragma unroll
        for (int i = 0; i < 8; ++i) xr2[i] = nxr[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) xr2[i] = ld4(xb, rl_off(rl, i));
      }
      lds_barrier();   // every wave is done with the weight slices (dX): their region takes the second x panel
#pragma unroll
      for (int i = 0; i < 8; ++i) rl_to_panels(rl, ownC, WIMG + wave * P_TILE, i, xr2[i]);
      lds_barrier();
      for (int tile = wave; tile < 6; tile += NT) {
        const int j = tile >> 1, rt = tile & 1;
        const char* gimg = j == 0 ? QIMG : (j == 1 ? DOIMG : SCRALL);
        const char* ximg = rt ? WIMG : DQACC;
        f32x16 acc = zero16();
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const char* xa = ximg + kt * P_TILE;
            const char* gb = gimg + kt * P_TILE;
            acc = mfma32x3(tr_pair(xa + lk.trn[s][0], xa + lk.trn[s][1]), tr_pair(xa + P_IMG + lk.trn[s][0], xa + P_IMG + lk.trn[s][1]),
                           tr_pair(gb + lk.trn[s][0], gb + lk.trn[s][1]), tr_pair(gb + P_IMG + lk.trn[s][0], gb + P_IMG + lk.trn[s][1]), acc);
          }
        }
        float* dst = p.dw_slab + (int64_t)b * (HID * 3 * HID) + HID * j + 32 * hd + r;
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) dst[(32 * rt + (tt & 3) + 8 * (tt >> 2) + 4 * h) * (3 * HID)] = acc[tt];
      }
      for (int k = threadIdx.x; k < 96; k += nthreads) {
        float acc = 0.f;
        for (int w = 0; w < NT; ++w) acc += sdb[w * 96 + k];
        p.db_slab[(int64_t)b * (3 * HID) + HID * (k >> 5) + 32 * hd + (k & 31)] = acc;
      }
      if (hd == 1) lds_barrier();   // the epilogue reuses the tiles the images are in
    } else if (hd == 0) {
      A32_REQUEST(1, false);
    }
  }

  A32_MARK(30);
  // ---- dx = dX + dz1 (residual), then back through the LayerNorm (and, for layer 0, the dropout) that produced x: row layout -----
  A32_LANE_CONSTS();
  A32_DZ_ROW();
  const DropCtx dce = b4r_drop_ctx(p.drop_e);
  f32x4 dxr[8];
  acc_to_rl(rl, ownA, ownB, r, h, dx, dxr);
  const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gprev + 4 * rl_c4);
  f32x4 dgam = {0.f, 0.f, 0.f, 0.f}, dbet = {0.f, 0.f, 0.f, 0.f};
  f32x4 dzo[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int trow = 32 * wave + 4 * i + rl_row;
    const bool lv = trow < L;
    const int64_t grow = row0 + min(trow, L - 1);
    f32x4 d4 = (dxr[i] + ld4(dab, rl_off(rl, i))) + dz_row(i);
    f32x4 zz;
    if (EMBED) {
      d4 = b4r_drop4(dce, d4, (uint64_t)(row0 + trow) * HID + (uint64_t)(4 * rl_c4));
      int64_t id = p.ids[grow];
      if (id < 0 || id >= p.V) id = 0;   // as the forward: out-of-range ids read the PAD row
      zz = *reinterpret_cast<const f32x4*>(p.table + id * HID + 4 * rl_c4) +
           *reinterpret_cast<const f32x4*>(p.pos + (int64_t)min(trow, L - 1) * HID + 4 * rl_c4);
    } else {
      zz = ld4(p.zprev + row0 * HID, rl_off(rl, i));
    }
    const float mean = (p.meanp + row0)[(uint32_t)min(trow, L - 1)], rstd = (p.rstdp + row0)[(uint32_t)min(trow, L - 1)];
    const f32x4 xhat = (zz - mean) * rstd;
    const f32x4 ge = d4 * gm;
    const float c1 = row_allsum16(sum4(ge)) * (1.0f / HID), c2 = row_allsum16(sum4(ge * xhat)) * (1.0f / HID);
    if (lv) { dgam += d4 * xhat; dbet += d4; }
#pragma unroll
    for (int e = 0; e < 4; ++e) dzo[i][e] = rstd * (ge[e] - c1 - xhat[e] * c2);
  }
  // column sums over the wave's tokens: the four lane rows hold different tokens of the same columns
  float* const myred = sred + wave * 128;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    dgam[e] += other_row(dgam[e]); dgam[e] += other_half(dgam[e], h);
    dbet[e] += other_row(dbet[e]); dbet[e] += other_half(dbet[e], h);
  }
  if (lane < 16) {
    *reinterpret_cast<f32x4*>(&myred[4 * rl_c4]) = dgam;
    *reinterpret_cast<f32x4*>(&myred[64 + 4 * rl_c4]) = dbet;
  }
  A32_MARK(31);
  lds_barrier();
  for (int k = threadIdx.x; k < 128; k += nthreads) {
    float acc = 0.f;
    for (int w = 0; w < NT; ++w) acc += sred[w * 128 + k];
    p.ln_part[(int64_t)b * 128 + k] = acc;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (32 * wave + 4 * i + rl_row < L) st4(dab, rl_off(rl, i), dzo[i]);
  A32_MARK(32);
}


// ---------------------------------------------------------------------------------------------------------------------------
// forward.  One workgroup per sequence, wave w owns tokens 32w .. 32w+31 end to end:
//   x rows (row layout, coalesced; for the first layer formed here from the embedding tables) -> images; Wqkv, Wo -> images
//   q~, k, v of BOTH heads for its tokens (transposed: rows = feature, lane = token), K / V rows -> the images every wave reads
//   per head: S^T[key][query] = K . Q~^T for all key tiles (the score row of a query lives in one lane's registers and its partner
//   half: softmax without LDS), dropout decisions from the counter hash (one hash per 4 keys) -> Pd, O^T = V^T . Pd^T
//   y^T = Wo^T . O^T, then in row layout: bias, dropout, residual, LayerNorm; ctx, z1 (, x1), statistics leave as whole rows.
// LDS: [Wo images 16 KB | region A 16 KB per 32 tokens: first Wqkv images 48 KB + the waves' x images, then K / V images of both
// heads, then the waves' transposition space | key mask adders, bqkv]
// ---------------------------------------------------------------------------------------------------------------------------
struct A32FwdP {
  const float* x; const int64_t* mask;
  const float* Wqkv; const float* bqkv; const float* Wo; const float* bo; const float* g1; const float* be1;
  float* qkv; float* ctx; float* lse; uint32_t* bits;
  float* z1; float* x1; float* mean1; float* rstd1;
  int B, L, NT;
  float qscale, eps;
  DropArgs drop_p, drop_o;
  const int64_t* ids; const float* table; const float* pos; const float* g0; const float* be0;
  float* x_out; float* mean0; float* rstd0;
  int V; float eps0;
  DropArgs drop_e;
  // CQ: the only rows anything downstream reads -- token clamp(slot_pos[b][j]), j < slots <= 64 (every slot, labelled or padded)
  const int64_t* slot_pos; int slots;
};

__host__ __device__ constexpr int fwd32_region_a(int NT) { return 4 * NT * P_TILE > 12 * P_TILE + 2 * NT * P_TILE ? 4 * NT * P_TILE : 12 * P_TILE + 2 * NT * P_TILE; }
__host__ __device__ constexpr int fwd32_lds(int NT) { return 4 * P_TILE + fwd32_region_a(NT) + (NT * 32 + 192) * 4; }
// CQ: + compact Q~ images [head][2 tiles], per-head (max, sum) of the key owners' partial softmax [2][NT][2][32][2], token -> slot [NT x 32],
// slot -> token [64]
__host__ __device__ constexpr int fwd32_lds_cq(int NT) { return fwd32_lds(NT) + 4 * P_TILE + (2 * NT * 2 * 64 + NT * 32 + 64) * 4; }

// CQ (the last layer under B4R_FLAG_HEAD_ROWS_ONLY, at most 64 slots on three or more token tiles): only the slots' rows of this block's
// outputs are read downstream (the feed-forward half in its slot mode, the backward's CQ form), so only those QUERIES are swept.  Every
// wave still projects q~ / k / v of its own tokens and writes its K / V images; the slot tokens scatter their q~ rows into compact Q~
// images; then every wave owns the 32 KEYS of its tokens for the one or two compact query tiles (S^T = K_w . Q~^T, a local softmax over
// its keys, an O^T partial over its own -- then dead -- K / V tile) and waves 0 .. NQ - 1 merge the NT partials of their tile (running
// maximum, rescaled sums), project, and run the row epilogue on the slots' rows: ctx, z1 (x1), statistics, lse and the decision words
// leave for the slots' TOKENS only.  Padded slots repeat position 0: one slot per token computes the row, the others are empty.
template <int NTT, bool DROP, bool CQ = false>
__global__ __launch_bounds__(512, 2) void attn32_fwd_kernel(A32FwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int NT = p.NT, L = p.L;
  char* const WOIMG = smem32;                       // [context half][hidden panel]: tile 2 ci + hj
  char* const RA = smem32 + 4 * P_TILE;
  char* const WQKV = RA;                            // phase 1: tile 2 cp + rt (column panel cp = 0..5 of [64][192], hidden rows 32 rt ..)
  char* const XIMG = RA + 12 * P_TILE;              // phase 1: wave w's x images: two column panels
  char* const KV = RA;                              // phase 2: [head][K | V][token tile]
  float* const sAdd = reinterpret_cast<float*>(RA + fwd32_region_a(NT));   // (mask adder - amax) * log2e per key, -inf beyond L
  float* const sbq = sAdd + NT * 32;
  char* const QC = reinterpret_cast<char*>(sbq + 192);                      // CQ: compact Q~ images, tile 2 hd + t
  float* const sMS = reinterpret_cast<float*>(QC + 4 * P_TILE);             // CQ: [head][key tile][compact tile][32 queries][max, sum]
  int* const sSlot = reinterpret_cast<int*>(sMS + 2 * NT * 2 * 64);        // CQ: [NT * 32] a slot of the token, -1: none
  int* const sTok = sSlot + NT * 32;                                        // CQ: [64] the slot's token, -1: empty (padded duplicate)
  const int NQ = CQ ? (p.slots + 31) >> 5 : 0;

  const int nthreads = blockDim.x;
  const int b = blockIdx.x;
  const int64_t row0 = (int64_t)b * L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  A32_LANE_CONSTS();
  char* const own = RA + wave * 4 * P_TILE;         // phase 3: 16 KB of transposition space per wave
  if (CQ) {   // token <-> slot (a token named by several slots -- position 0 of padded slots -- keeps one of them)
    if ((int)threadIdx.x < NT * 32) sSlot[threadIdx.x] = -1;
    lds_barrier();
    if ((int)threadIdx.x < p.slots) {
      const int64_t q = p.slot_pos[(int64_t)b * p.slots + threadIdx.x];
      sSlot[q < 0 ? 0 : (q >= L ? L - 1 : (int)q)] = threadIdx.x;
    }
    lds_barrier();
    if (threadIdx.x < 64) {
      int tq = -1;
      if ((int)threadIdx.x < p.slots) {
        const int64_t q = p.slot_pos[(int64_t)b * p.slots + threadIdx.x];
        const int qc = q < 0 ? 0 : (q >= L ? L - 1 : (int)q);
        tq = sSlot[qc] == (int)threadIdx.x ? qc : -1;
      }
      sTok[threadIdx.x] = tq;
    }
  }
  // (s_setprio(1) for waves 4.., the younger wave of every SIMD, was measured: forward 42.7 / 42.9 us with, 43.0 / 42.2 without)

  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  for (int k = threadIdx.x; k < 192; k += nthreads) sbq[k] = p.bqkv[k];

  A32F_MARK(0);
  // ---- the wave's x rows (row layout).  First layer: x = dropout(LayerNorm(table[id] + position)), written out with its statistics
  const float* const xb = (p.ids != nullptr ? p.x_out : p.x) + row0 * HID;
  f32x4 xr[8];
  if (p.ids != nullptr) {
    const DropCtx dce = b4r_drop_ctx(p.drop_e);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(p.g0 + 4 * rl_c4), be = *reinterpret_cast<const f32x4*>(p.be0 + 4 * rl_c4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int trow = 32 * wave + 4 * i + rl_row, trc = min(trow, L - 1);
      int64_t id = p.ids[row0 + trc];
      if (id < 0 || id >= p.V) id = 0;   // out-of-range ids read the PAD row, as b4r_embed_ln_fwd does
      const f32x4 e = *reinterpret_cast<const f32x4*>(p.table + id * HID + 4 * rl_c4) +
                      *reinterpret_cast<const f32x4*>(p.pos + (int64_t)trc * HID + 4 * rl_c4);
      const float mean = row_allsum16(sum4(e)) * (1.0f / HID);
      const f32x4 d = e - mean;
      const float rstd = rsqrtf(row_allsum16(sum4(d * d)) * (1.0f / HID) + p.eps0);
      f32x4 y;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float inv = rstd * gm[c];
        y[c] = e[c] * inv + (be[c] - mean * inv);
      }
      y = b4r_drop4(dce, y, (uint64_t)(row0 + trow) * HID + (uint64_t)(4 * rl_c4));
      xr[i] = y;
      if (trow < L) {
        st4(p.x_out + row0 * HID, rl_off(rl, i), y);
        if (rl_c4 == 0) {
          if (p.mean0) p.mean0[row0 + trow] = mean;
          if (p.rstd0) p.rstd0[row0 + trow] = rstd;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) xr[i] = ld4(xb, rl_off(rl, i));
  }
  A32F_MARK(1);
  // weights: Wqkv [64][192] as 6 column panels x 2 row tiles, Wo [64][64] as tile 2 ci + hj (context rows 32 ci .., hidden 32 hj ..)
  stage_tiles<5>(WOIMG, 16, nthreads, [&](int gt, int& ld) __attribute__((always_inline)) -> const float* {
    const bool wo = gt < 4;
    const int g = wo ? gt : gt - 4;
    ld = wo ? HID : 3 * HID;
    return wo ? p.Wo + (int64_t)32 * (g >> 1) * HID + 32 * (g & 1) : p.Wqkv + (int64_t)32 * (g & 1) * (3 * HID) + 32 * (g >> 1);
  });
#pragma unroll
  for (int i = 0; i < 8; ++i) rl_to_panels(rl, XIMG + wave * 2 * P_TILE, XIMG + wave * 2 * P_TILE + P_TILE, i, xr[i]);
  A32F_MARK(2);
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;   // (barrier: the weight images and sbq are in place)
  A32F_MARK(3);
  if ((int)threadIdx.x < NT * 32)
    sAdd[threadIdx.x] = (int)threadIdx.x < L ? (((1.0f - (float)mval) * -1e9f) - amax) * LOG2E : -INFINITY;

  // ---- q~, k, v of both heads (transposed), the wave's tokens ------------------------------------------------------------------
  bf16x8 qBh[2][2], qBl[2][2];     // [head][k-step]: Q~^T[feature][query] as the B operand of S^T = K . Q~^T
  f32x16 kT[2], vT[2];
  {
    bf16x8 xh[4], xl[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const char* xt = XIMG + (wave * 2 + (ks >> 1)) * P_TILE + lk.rowc[ks & 1];
      xh[ks] = row_at(xt); xl[ks] = row_at(xt + P_IMG);
    }
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
      f32x16 acc[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[j] = rows_of(sbq + HID * j + 32 * hd, h);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const char* wt = WQKV + (2 * (2 * j + hd) + (ks >> 1)) * P_TILE;
          const int a0 = lk.trn[ks & 1][0], a1 = lk.trn[ks & 1][1];
          acc[j] = mfma32x3(tr_pair(wt + a0, wt + a1), tr_pair(wt + P_IMG + a0, wt + P_IMG + a1), xh[ks], xl[ks], acc[j]);
        }
      }
      acc[0] = acc[0] * p.qscale;
      if (p.qkv != nullptr && live) {   // b4r_attn_fwd's layout (compatibility output; the train step does not ask for it)
        float* dst = p.qkv + (row0 + tok) * (3 * HID) + 32 * hd + 4 * h;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int gp = 0; gp < 4; ++gp)
            *reinterpret_cast<f32x4*>(dst + HID * j + 8 * gp) = (f32x4){acc[j][4 * gp], acc[j][4 * gp + 1], acc[j][4 * gp + 2], acc[j][4 * gp + 3]};
      }
      acc[0] = acc[0] * LOG2E;
      if (CQ) {   // (every key masked: zero, as the dense form's operand)
        const int js_ = live ? sSlot[tok] : -1;
        if (js_ >= 0) acc_rows_at(QC + (2 * hd + (js_ >> 5)) * P_TILE, js_ & 31, h, acc[0] * (amax != 0.0f ? 0.0f : 1.0f));
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) acc_frag(acc[0], s, qBh[hd][s], qBl[hd][s]);
      }
      kT[hd] = acc[1]; vT[hd] = acc[2];
    }
  }
  if (amax != 0.0f) {   // every key masked: Keras' -1e9 absorbs the scores -> a uniform softmax over the L keys (lse = log L)
#pragma unroll
    for (int hd = 0; hd < 2; ++hd)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) { qBh[hd][s][e] = (__bf16)0.f; qBl[hd][s][e] = (__bf16)0.f; }
  }
  A32F_MARK(4);
  lds_barrier();   // every wave is done with the Wqkv and x images: region A takes the K / V images
  A32F_MARK(5);
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    f32x16 kk = kT[hd], vv = vT[hd];
    if (!live) { kk = zero16(); vv = zero16(); }   // pad tokens: zero rows (their keys are masked anyway; no NaN may enter P . V)
    acc_to_rows(KV + ((2 * hd) * NT + wave) * P_TILE, lk, kk);
    acc_to_rows(KV + ((2 * hd + 1) * NT + wave) * P_TILE, lk, vv);
  }
  A32F_MARK(6);
  lds_barrier();
  A32F_MARK(7);

  // ---- attention, one head at a time --------------------------------------------------------------------------------------
  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  f32x16 O[2];
  const int qt = wave;
  const int slot = (r & 24) | ((r & 3) << 1) | ((r >> 2) & 1);   // query 16s + 8a + 4h' + b -> 16s + 8a + 2b + h' (the backward's register pairs)
  if (CQ) {
    const f32x16 add = rows_of(sAdd + 32 * wave, h);
    const bool qwave = wave < NQ;
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
      char* const kown = KV + ((2 * hd) * NT + wave) * P_TILE;
      char* const vown = KV + ((2 * hd + 1) * NT + wave) * P_TILE;
      const int64_t bh = (int64_t)b * 2 + hd;
      float* const ms_h = sMS + hd * (NT * 2 * 64);
      bf16x8 kAh[2], kAl[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kAh[ks] = row_at(kown + lk.rowc[ks]); kAl[ks] = row_at(kown + P_IMG + lk.rowc[ks]); }
      f32x16 Op[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        Op[t] = zero16();
        if (t >= NQ) continue;   // (block-uniform)
        const int tq = sTok[32 * t + r];   // this lane's query: the token of slot 32 t + r, -1: empty
        f32x16 S = add;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const char* qi = QC + (2 * hd + t) * P_TILE + lk.rowc[ks];
          S = mfma32x3(kAh[ks], kAl[ks], row_at(qi), row_at(qi + P_IMG), S);   // S^T[key][query] = K_w . Q~_t^T
        }
        float m = S[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) m = fmaxf(m, S[e]);
        m = fmaxf(m, other_half(m, h));
        const float mref = m == -INFINITY ? 0.0f : m;   // a tile of pad keys only: every term is 0
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { S[e] = __builtin_amdgcn_exp2f(S[e] - mref); sum += S[e]; }
        sum += other_half(sum, h);
        if (h == 0) { float* ms = ms_h + ((wave * 2 + t) * 32 + r) * 2; ms[0] = m; ms[1] = sum; }
        if (DROP) {
          uint32_t word = 0;
          const uint64_t dbase = ((uint64_t)bh * L + (uint64_t)max(tq, 0)) * (uint64_t)B4R_ATTN_PITCH;
#pragma unroll
          for (int gp = 0; gp < 4; ++gp) {
            const B4rKeep4 k4 = b4r_keep4p(dcp, dbase + (uint64_t)(32 * wave + 8 * gp + 4 * h));
#pragma unroll
            for (int e = 0; e < 4; ++e) S[4 * gp + e] = k4.k[e] ? S[4 * gp + e] : 0.f;
            word |= k4.bits() << (8 * gp + 4 * h);
          }
          word |= other_half_u(word, h);
          if (h == 0 && tq >= 0) p.bits[((bh * NT + wave) * NT + (tq >> 5)) * 32 + a32_slot_perm(tq & 31)] = word;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 ph, pl;
          acc_frag(S, s2, ph, pl);
          Op[t] = mfma32x3(tr_pair(vown + lk.trp[s2][0], vown + lk.trp[s2][1]), tr_pair(vown + P_IMG + lk.trp[s2][0], vown + P_IMG + lk.trp[s2][1]),
                           ph, pl, Op[t]);
        }
      }
      // the partials go over the wave's own K / V tile of this head (its reads of them above are the wave's own, consumed by products)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t >= NQ) continue;
        char* dst = (t == 0 ? kown : vown) + lane * 16;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          *reinterpret_cast<f32x4*>(dst + jj * 1024) = (f32x4){Op[t][4 * jj], Op[t][4 * jj + 1], Op[t][4 * jj + 2], Op[t][4 * jj + 3]};
      }
      lds_barrier();
      O[hd] = zero16();
      if (qwave) {   // merge of compact tile t = wave
        const int tq = sTok[32 * wave + r];
        float M = -INFINITY;
        for (int w_ = 0; w_ < NT; ++w_) M = fmaxf(M, ms_h[((w_ * 2 + wave) * 32 + r) * 2]);
        float tot = 0.f;
        for (int w_ = 0; w_ < NT; ++w_) {
          const float* ms = ms_h + ((w_ * 2 + wave) * 32 + r) * 2;
          const float a = ms[0] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ms[0] - M);
          tot = fmaf(a, ms[1], tot);
          const char* srcp = KV + ((2 * hd + (wave == 0 ? 0 : 1)) * NT + w_) * P_TILE + lane * 16;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x4 o4 = *reinterpret_cast<const f32x4*>(srcp + jj * 1024);
#pragma unroll
            for (int e = 0; e < 4; ++e) O[hd][4 * jj + e] = fmaf(a, o4[e], O[hd][4 * jj + e]);
          }
        }
        const float inv = (DROP ? dcp.scale : 1.0f) / tot;
        O[hd] = O[hd] * inv;
        if (h == 0 && tq >= 0 && p.lse) p.lse[bh * L + tq] = M * LN2 + __logf(tot);
      }
    }
  } else {
#pragma unroll
  for (int hd = 0; hd < 2; ++hd) {
    const char* kimg = KV + (2 * hd) * NT * P_TILE;
    const char* vimg = KV + (2 * hd + 1) * NT * P_TILE;
    const int64_t bh = (int64_t)b * 2 + hd;
    f32x16 S[NTT];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
      if (t < NT) {   // (wave-uniform; NTT is the compile-time bound of the token tiles)
        S[t] = rows_of(sAdd + 32 * t, h);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const char* ka = kimg + t * P_TILE + lk.rowc[ks];
          S[t] = mfma32x3(row_at(ka), row_at(ka + P_IMG), qBh[hd][ks], qBl[hd][ks], S[t]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) S[t][e] = -INFINITY;
      }
    }
    A32F_MARK(8 + 4 * hd);
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, S[t][e]);
    m = fmaxf(m, other_half(m, h));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NTT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) { S[t][e] = __builtin_amdgcn_exp2f(S[t][e] - m); sum += S[t][e]; }
    sum += other_half(sum, h);
    const float inv = 1.0f / sum;
    if (h == 0 && live && p.lse) p.lse[bh * L + tok] = m * LN2 + __logf(sum);
    A32F_MARK(9 + 4 * hd);
    O[hd] = zero16();
    const uint64_t dbase = ((uint64_t)bh * L + (uint64_t)(live ? tok : 0)) * (uint64_t)B4R_ATTN_PITCH;
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
      if (t >= NT) continue;
      if (DROP) {
        uint32_t word = 0;
        const float sc = inv * dcp.scale;
#pragma unroll
        for (int gp = 0; gp < 4; ++gp) {
          const B4rKeep4 k4 = b4r_keep4p(dcp, dbase + (uint64_t)(32 * t + 8 * gp + 4 * h));
#pragma unroll
          for (int e = 0; e < 4; ++e) S[t][4 * gp + e] = k4.k[e] ? S[t][4 * gp + e] * sc : 0.f;
          word |= k4.bits() << (8 * gp + 4 * h);
        }
        word |= other_half_u(word, h);
        if (h == 0 && live) p.bits[((bh * NT + t) * NT + qt) * 32 + slot] = word;
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) S[t][e] *= inv;
      }
      // O^T[feature][query] += V^T[feature][keys of tile t] . Pd^T[key][query]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ph, pl;
        acc_frag(S[t], s2, ph, pl);
        const char* va = vimg + t * P_TILE;
        O[hd] = mfma32x3(tr_pair(va + lk.trp[s2][0], va + lk.trp[s2][1]), tr_pair(va + P_IMG + lk.trp[s2][0], va + P_IMG + lk.trp[s2][1]),
                         ph, pl, O[hd]);
      }
    }
  }

  }
  A32F_MARK(16);
  // ---- y^T[hidden][token] = Wo^T . ctx^T: registers 8s .. 8s+7 of O are context columns 16s + 8h + .. of the head: natural k order
  f32x16 y[2] = {zero16(), zero16()};
#pragma unroll
  for (int hd = 0; hd < 2; ++hd)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 oh, ol;
      acc_frag(O[hd], s, oh, ol);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const char* wt = WOIMG + (2 * hd + rt) * P_TILE;
        y[rt] = mfma32x3(tr_pair(wt + lk.trn[s][0], wt + lk.trn[s][1]), tr_pair(wt + P_IMG + lk.trn[s][0], wt + P_IMG + lk.trn[s][1]), oh, ol, y[rt]);
      }
    }
  A32F_MARK(17);
  lds_barrier();   // every wave is done with the K / V images: the waves' transposition space overlays them
  A32F_MARK(18);

  if (CQ) {   // the rows of the wave's compact tile: slot 32 wave + 4 i + rl_row -> its token's row (empty slots: nothing)
    if (wave >= NQ) return;
    f32x4 cr[8], yr[8];
    acc_to_rl<true>(rl, own, own + P_TILE, r, h, O, cr);
    acc_to_rl(rl, own + 2 * P_TILE, own + 3 * P_TILE, r, h, y, yr);
    const DropCtx dco = b4r_drop_ctx(p.drop_o);
    const f32x4 bo4 = *reinterpret_cast<const f32x4*>(p.bo + 4 * rl_c4);
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.g1 + 4 * rl_c4), be4 = *reinterpret_cast<const f32x4*>(p.be1 + 4 * rl_c4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tj = sTok[32 * wave + 4 * i + rl_row];
      const uint32_t off = (uint32_t)(max(tj, 0) * HID + 4 * rl_c4);
      const f32x4 res = ld4(xb, off);
      const f32x4 z = res + b4r_drop4(dco, yr[i] + bo4, (uint64_t)(row0 + max(tj, 0)) * HID + (uint64_t)(4 * rl_c4));
      const float mean = row_allsum16(sum4(z)) * (1.0f / HID);
      const f32x4 d = z - mean;
      const float rstd = rsqrtf(row_allsum16(sum4(d * d)) * (1.0f / HID) + p.eps);
      if (tj >= 0) {
        if (p.ctx) st4(p.ctx + row0 * HID, off, cr[i]);
        if (p.z1) st4(p.z1 + row0 * HID, off, z);
        if (p.x1) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float inv = rstd * g4[c];
            o[c] = fmaf(z[c], inv, fmaf(-mean, inv, be4[c]));
          }
          st4(p.x1 + row0 * HID, off, o);
        }
        if (rl_c4 == 0) {
          if (p.mean1) p.mean1[row0 + tj] = mean;
          if (p.rstd1) p.rstd1[row0 + tj] = rstd;
        }
      }
    }
    return;
  }
  // ---- row layout: ctx out; z1 = x + dropout(y + bo), LayerNorm ------------------------------------------------------------------
  {
    f32x4 cr[8];
    acc_to_rl<true>(rl, own, own + P_TILE, r, h, O, cr);
    if (p.ctx) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (32 * wave + 4 * i + rl_row < L) st4(p.ctx + row0 * HID, rl_off(rl, i), cr[i]);
    }
  }
  A32F_MARK(19);
  f32x4 yr[8];
  acc_to_rl(rl, own + 2 * P_TILE, own + 3 * P_TILE, r, h, y, yr);   // (the second 8 KB of the wave's space: no wait for the ctx reads)
  const DropCtx dco = b4r_drop_ctx(p.drop_o);
  const f32x4 bo4 = *reinterpret_cast<const f32x4*>(p.bo + 4 * rl_c4);
  const f32x4 g4 = *reinterpret_cast<const f32x4*>(p.g1 + 4 * rl_c4), be4 = *reinterpret_cast<const f32x4*>(p.be1 + 4 * rl_c4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int trow = 32 * wave + 4 * i + rl_row;
    const f32x4 res = ld4(xb, rl_off(rl, i));
    const f32x4 z = res + b4r_drop4(dco, yr[i] + bo4, (uint64_t)(row0 + trow) * HID + (uint64_t)(4 * rl_c4));
    const float mean = row_allsum16(sum4(z)) * (1.0f / HID);
    const f32x4 d = z - mean;
    const float rstd = rsqrtf(row_allsum16(sum4(d * d)) * (1.0f / HID) + p.eps);
    if (trow < L) {
      if (p.z1) st4(p.z1 + row0 * HID, rl_off(rl, i), z);
      if (p.x1) {
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float inv = rstd * g4[c];
          o[c] = fmaf(z[c], inv, fmaf(-mean, inv, be4[c]));   // x1_load4's formula, rounding for rounding
        }
        st4(p.x1 + row0 * HID, rl_off(rl, i), o);
      }
      if (rl_c4 == 0) {
        if (p.mean1) p.mean1[row0 + trow] = mean;
        if (p.rstd1) p.rstd1[row0 + trow] = rstd;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The attention CORE alone on 32-token tiles, for every hidden size with 32-wide heads (H = 128: 4 heads, H = 256: 8 heads; the
// layers around it are tile products there): one workgroup per (sequence, head), q | k | v read from the [N, 3H] projection output
// (q pre-scaled by 1/sqrt(d): b4r_attn_fwd's contract), the same score / softmax / dropout / sweep code as the resident kernels above
// -- K and V images in LDS, the score row of a query in one lane pair, dK / dV in registers, the dQ partials ordered by flags.
// Replaces round 1's attn_rx_fwd / attn_rx_dq / attn_rx_dkv kernels (16-token tiles, three launches) where L <= 224.
// ---------------------------------------------------------------------------------------------------------------------------
struct A32CoreFwdP {
  const float* qkv; const int64_t* mask; float* ctx; float* lse; uint32_t* bits;
  int B, L, NT, heads;
  DropArgs drop_p;
};
__host__ __device__ constexpr int core_fwd_lds(int NT) { return 2 * NT * P_TILE + NT * 32 * 4; }

// the wave's rows of one 32-column slice of a [N, ld] tensor in ACCUMULATOR layout: register 4 gp + e of lane (r, h) = column
// 8 gp + 4h + e of token r (what a product with the tokens on the lanes leaves: acc_frag / acc_to_rows take it from there)
__device__ __forceinline__ f32x16 load_acc_layout(const float* row_col0, int h) {
  f32x16 v;
#pragma unroll
  for (int gp = 0; gp < 4; ++gp) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(row_col0 + 8 * gp + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * gp + e] = q[e];
  }
  return v;
}

template <int NTT, bool DROP>
__global__ __launch_bounds__(512, 2) void attn32_core_fwd_kernel(A32CoreFwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int NT = p.NT, L = p.L;
  char* const kimg = smem32;                          // K images, one tile per 32 tokens
  char* const vimg = smem32 + NT * P_TILE;            // V images
  float* const sAdd = reinterpret_cast<float*>(smem32 + 2 * NT * P_TILE);   // (mask adder - amax) * log2e per key, -inf beyond L
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  const int Hh = 32 * p.heads;
  const int64_t row0 = (int64_t)b * L;
  const int64_t bh = (int64_t)b * p.heads + head;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  A32_LANE_CONSTS();
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  // the wave's tokens: q~ = q . log2(e) as the B operand of S^T = K . Q~^T, k and v rows -> the images every wave reads
  const float* const src = p.qkv + (row0 + tokc) * (3 * Hh) + 32 * head;
  f32x16 qa = load_acc_layout(src, h), kacc = load_acc_layout(src + Hh, h), vacc = load_acc_layout(src + 2 * Hh, h);
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
  if ((int)threadIdx.x < NT * 32)
    sAdd[threadIdx.x] = (int)threadIdx.x < L ? (((1.0f - (float)mval) * -1e9f) - amax) * LOG2E : -INFINITY;
  qa = qa * (amax != 0.0f ? 0.0f : LOG2E);   // every key masked: Keras' -1e9 absorbs the scores -> a uniform softmax over the L keys
  bf16x8 qBh[2], qBl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) acc_frag(qa, s, qBh[s], qBl[s]);
  if (!live) { kacc = zero16(); vacc = zero16(); }   // pad tokens: zero rows (their keys are masked anyway; no NaN may enter P . V)
  acc_to_rows(kimg + wave * P_TILE, lk, kacc);
  acc_to_rows(vimg + wave * P_TILE, lk, vacc);
  lds_barrier();

  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  f32x16 O;
  const int qt = wave;
  const int slot = (r & 24) | ((r & 3) << 1) | ((r >> 2) & 1);   // query 16s + 8a + 4h' + b -> 16s + 8a + 2b + h' (the backward's register pairs)
  {
  f32x16 S[NTT];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
    if (t < NT) {   // (wave-uniform; NTT is the compile-time bound of the token tiles)
      S[t] = rows_of(sAdd + 32 * t, h);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char* ka = kimg + t * P_TILE + lk.rowc[ks];
        S[t] = mfma32x3(row_at(ka), row_at(ka + P_IMG), qBh[ks], qBl[ks], S[t]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) S[t][e] = -INFINITY;
    }
  }
  
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) m = fmaxf(m, S[t][e]);
  m = fmaxf(m, other_half(m, h));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NTT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[t][e] = __builtin_amdgcn_exp2f(S[t][e] - m); sum += S[t][e]; }
  sum += other_half(sum, h);
  const float inv = 1.0f / sum;
  if (h == 0 && live && p.lse) p.lse[bh * L + tok] = m * LN2 + __logf(sum);
  
  O = zero16();
  const uint64_t dbase = ((uint64_t)bh * L + (uint64_t)(live ? tok : 0)) * (uint64_t)B4R_ATTN_PITCH;
#pragma unroll
  for (int t = 0; t < NTT; ++t) {
    if (t >= NT) continue;
    if (DROP) {
      uint32_t word = 0;
      const float sc = inv * dcp.scale;
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {
        const B4rKeep4 k4 = b4r_keep4p(dcp, dbase + (uint64_t)(32 * t + 8 * gp + 4 * h));
#pragma unroll
        for (int e = 0; e < 4; ++e) S[t][4 * gp + e] = k4.k[e] ? S[t][4 * gp + e] * sc : 0.f;
        word |= k4.bits() << (8 * gp + 4 * h);
      }
      word |= other_half_u(word, h);
      if (h == 0 && live) p.bits[((bh * NT + t) * NT + qt) * 32 + slot] = word;
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) S[t][e] *= inv;
    }
    // O^T[feature][query] += V^T[feature][keys of tile t] . Pd^T[key][query]
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 ph, pl;
      acc_frag(S[t], s2, ph, pl);
      const char* va = vimg + t * P_TILE;
      O = mfma32x3(tr_pair(va + lk.trp[s2][0], va + lk.trp[s2][1]), tr_pair(va + P_IMG + lk.trp[s2][0], va + P_IMG + lk.trp[s2][1]),
                       ph, pl, O);
    }
  }
  }
  // O's registers 8s .. 8s+7 are context columns 16s + 8h + (0..7) of the head (V image in swapped column order, transposed reads)
  if (live) {
    float* dst = p.ctx + (row0 + tok) * Hh + 32 * head + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<f32x4*>(dst + 16 * s) = (f32x4){O[8 * s], O[8 * s + 1], O[8 * s + 2], O[8 * s + 3]};
      *reinterpret_cast<f32x4*>(dst + 16 * s + 4) = (f32x4){O[8 * s + 4], O[8 * s + 5], O[8 * s + 6], O[8 * s + 7]};
    }
  }
}

struct A32CoreBwdP {
  const float* qkv; const float* dctx; const float* ctx; const float* lse; const uint32_t* bits; const int64_t* mask;
  float* dqkv;
  int B, L, NT, heads;
  float qscale;
  DropArgs drop_p;
};
// LDS (bytes): [Q~ images NT x 4 KB | dO images | dQ accumulators (fp32, register layout) | per-wave scratch | mask adders, -lse, D, flags]
__host__ __device__ constexpr int core_bwd_lds(int NT) { return 4 * NT * P_TILE + (3 * NT * 32 + 8) * 4; }

template <bool DROP>
__global__ __launch_bounds__(512, 2) void attn32_core_bwd_kernel(A32CoreBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int NT = p.NT, L = p.L;
  char* const QIMG = smem32;
  char* const DOIMG = QIMG + NT * P_TILE;
  char* const DQACC = DOIMG + NT * P_TILE;
  char* const SCRALL = DQACC + NT * P_TILE;
  float* const sAdd = reinterpret_cast<float*>(SCRALL + NT * P_TILE);
  float* const sCS = sAdd + NT * 32;                  // -lse * log2e per query (-inf: pad)
  float* const sD = sCS + NT * 32;                    // rowsum(dctx * ctx) per query
  int* const sflag = reinterpret_cast<int*>(sD + NT * 32);
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  const int Hh = 32 * p.heads;
  const int64_t row0 = (int64_t)b * L;
  const int64_t bh = (int64_t)b * p.heads + head;
  constexpr int hd = 0; (void)hd;                     // (the stamp macros of the resident kernel name a head)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const scr = SCRALL + wave * P_TILE;
  char* const ownA = QIMG + wave * P_TILE;
  char* const ownB = DOIMG + wave * P_TILE;
  char* const ownC = DQACC + wave * P_TILE;
  A32_LANE_CONSTS();
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  // the wave's tokens in accumulator layout: q~, k, v of the head, dctx and ctx columns of the head, the query's lse
  const float* const src = p.qkv + (row0 + tokc) * (3 * Hh) + 32 * head;
  f32x16 qT = load_acc_layout(src, h), kT = load_acc_layout(src + Hh, h), vT = load_acc_layout(src + 2 * Hh, h);
  const f32x16 dcT = load_acc_layout(p.dctx + (row0 + tokc) * Hh + 32 * head, h);
  const f32x16 cxT = load_acc_layout(p.ctx + (row0 + tokc) * Hh + 32 * head, h);
  const float lse_q = live ? p.lse[bh * L + tok] : INFINITY;
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
  const bool dead = amax != 0.0f;   // every key masked: Keras' -1e9 absorbs the scores, the softmax is uniform over all L keys
  if ((int)threadIdx.x < NT * 32)
    sAdd[threadIdx.x] = (int)threadIdx.x < L ? (((1.0f - (float)mval) * -1e9f) - amax) * LOG2E : -INFINITY;
  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  const float pscale = DROP ? dcp.scale : 1.0f;
  qT = qT * LOG2E;
  {   // D = sum_c dctx * ctx over the head's 32 columns (the softmax backward's row term)
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) d = fmaf(dcT[e], cxT[e], d);
    d += other_half(d, h);
    if (h == 0) { sD[tok] = d; sCS[tok] = -lse_q * LOG2E; }
  }
  bf16x8 kBh[2], kBl[2], vBh[2], vBl[2];     // K^T, V^T [feature][key] as B operands of S = Q~.K^T and dA = dO.V^T
#pragma unroll
  for (int s = 0; s < 2; ++s) { acc_frag(kT, s, kBh[s], kBl[s]); acc_frag(vT, s, vBh[s], vBl[s]); }
  acc_to_rows(ownA, lk, qT);
  acc_to_rows(ownB, lk, dcT);
  acc_to_rows(scr, lk, kT);
  lds_barrier();                             // (the wave's own K^T image is written: its transposed reads below may start)
  bf16x8 kTh[2], kTl[2];                     // K^T[feature position][key] as the A operand of dQ^T = K^T.dS^T
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    kTh[s] = tr_pair(scr + lk.trn[s][0], scr + lk.trn[s][1]);
    kTl[s] = tr_pair(scr + P_IMG + lk.trn[s][0], scr + P_IMG + lk.trn[s][1]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(ownC + j * 1024 + lane * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (lane == 0) sflag[wave] = 0;
  lds_barrier();   // images, D, -lse of every tile in place; accumulators zero; (the K^T reads above have landed: lgkmcnt(0))

  f32x16 dK = zero16(), dV = zero16(), S, dA;
  const float adk = sAdd[tok];
  if (dead) {   // S = -lse for every key: the scores are absorbed by the mask adder
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { kBh[ks][e] = (__bf16)0.f; kBl[ks][e] = (__bf16)0.f; }
  }
#define A32_LD_TR(img, s2, H, Lo)                                                \
H = tr_pair((img) + lk.trp[s2][0], (img) + lk.trp[s2][1]);                     \
Lo = tr_pair((img) + P_IMG + lk.trp[s2][0], (img) + P_IMG + lk.trp[s2][1])
#define A32_LD_ROW(img, ks, H, Lo)                                               \
H = row_at((img) + lk.rowc[ks]);                                               \
Lo = row_at((img) + P_IMG + lk.rowc[ks])
#define A32_LD_SCR(ks, H, Lo)                                                    \
H = tr_pair(scr + lk.trn[ks][0], scr + lk.trn[ks][1]);                         \
Lo = tr_pair(scr + P_IMG + lk.trn[ks][0], scr + P_IMG + lk.trn[ks][1])
#define A32_SB() __builtin_amdgcn_sched_barrier(0)
  auto tile_of = [&](int s) __attribute__((always_inline)) { const int t = wave + s; return t >= NT ? t - NT : t; };
  typedef const __attribute__((address_space(4))) uint64_t* kmask_ptr;   // constant address space: scalar loads
  {   // S, dA of the first tile
    const char* qimg = QIMG + wave * P_TILE;
    const char* dimg = DOIMG + wave * P_TILE;
    S = rows_of(sCS + 32 * wave, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] += adk;
    dA = zero16();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah, al;
      A32_LD_ROW(qimg, ks, ah, al);
      S = mfma32x3(ah, al, kBh[ks], kBl[ks], S);
      A32_LD_ROW(dimg, ks, ah, al);
      dA = mfma32x3(ah, al, vBh[ks], vBl[ks], dA);
    }
  }
  for (int s = 0; s < NT; ++s) {
    const int t = tile_of(s), tn = tile_of(s + 1);   // (the last step forms S / dA of a tile nobody uses: no branch in the body)
    const char* qimg = QIMG + t * P_TILE;
    const char* dimg = DOIMG + t * P_TILE;
    const char* qn = QIMG + tn * P_TILE;
    const char* dn = DOIMG + tn * P_TILE;
    A32_SWEEP(2 + 4 * s);
    // ---- vector phase ------------------------------------------------------------------------------------------------------
    bf16x8 f0h, f0l, f1h, f1l;
    A32_LD_TR(dimg, 0, f0h, f0l);   // the first two products' fragments travel during the vector phase
    A32_LD_TR(qimg, 0, f1h, f1l);
    const f32x16 Dq = rows_of(sD + 32 * t, h);
    uint64_t km[16];   // keep decisions of the 32 x 32 block: register tt's lane mask over the keys
    if (DROP) {
      kmask_ptr mp = (kmask_ptr)(reinterpret_cast<const uint64_t*>(p.bits) + ((bh * NT + wave) * NT + t) * 16);
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) km[tt] = mp[tt];
    }
    bf16x8 pdh[2], pdl[2], dsh[2], dsl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f32x8 pd, ds;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tt = 8 * s2 + j;
        const float pr = (A32_EXP & 1) ? S[tt] : __builtin_amdgcn_exp2f(S[tt]);
        if (A32_EXP & 1) { pd[j] = pr; ds[j] = dA[tt]; } else
        if (DROP) {
          const float kf = __builtin_amdgcn_inverse_ballot_w64(km[tt]) ? pscale : 0.f;
          pd[j] = pr * kf;
          ds[j] = pr * fmaf(dA[tt], kf, -Dq[tt]);
        } else {
          pd[j] = pr;
          ds[j] = pr * (dA[tt] - Dq[tt]);
        }
      }
      split8(pd, pdh[s2], pdl[s2]);
      split8(ds, dsh[s2], dsl[s2]);
      // dS -> the wave's [key][query] scratch image: registers 4a .. 4a+3 of this half are queries 16 s2 + 8a + 4h + (0..3)
      const s16x8 hv = __builtin_bit_cast(s16x8, dsh[s2]), lv = __builtin_bit_cast(s16x8, dsl[s2]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        char* w8 = scr + p_chunk(r, 2 * s2 + a) + 8 * h;
        *reinterpret_cast<s16x4*>(w8) = a ? __builtin_shufflevector(hv, hv, 4, 5, 6, 7) : __builtin_shufflevector(hv, hv, 0, 1, 2, 3);
        *reinterpret_cast<s16x4*>(w8 + P_IMG) = a ? __builtin_shufflevector(lv, lv, 4, 5, 6, 7) : __builtin_shufflevector(lv, lv, 0, 1, 2, 3);
      }
    }
    // S of the next tile starts from -lse (per query) + the key's mask adder
    S = rows_of(sCS + 32 * tn, h);
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] += adk;
    A32_SWEEP(3 + 4 * s);
    A32_SB();
    // ---- matrix phase: ten products, fragments requested two products ahead ---------------------------------------------------
    bf16x8 f2h, f2l, f3h, f3l, f4h, f4l, f5h, f5l, f6h, f6l, f7h, f7l, f8h, f8l, f9h, f9l;
    A32_LD_TR(dimg, 1, f2h, f2l);
    dV = mfma32x3(f0h, f0l, pdh[0], pdl[0], dV);   // dV^T[feature][key] += dO^T[feature][query] . Pd[query][key], queries 0..15
    A32_SB();
    A32_LD_TR(qimg, 1, f3h, f3l);
    dK = mfma32x3(f1h, f1l, dsh[0], dsl[0], dK);   // dK^T += Q~^T . dS
    A32_SB();
    A32_LD_ROW(qn, 0, f4h, f4l);
    dV = mfma32x3(f2h, f2l, pdh[1], pdl[1], dV);
    A32_SB();
    A32_LD_ROW(dn, 0, f5h, f5l);
    dK = mfma32x3(f3h, f3l, dsh[1], dsl[1], dK);
    A32_SB();
    A32_LD_ROW(qn, 1, f6h, f6l);
    S = mfma32x3(f4h, f4l, kBh[0], kBl[0], S);     // S[query][key] = Q~ . K^T of the next tile
    A32_SB();
    A32_LD_ROW(dn, 1, f7h, f7l);
    dA = mfma32x3(f5h, f5l, vBh[0], vBl[0], zero16());
    A32_SB();
    A32_LD_SCR(0, f8h, f8l);
    S = mfma32x3(f6h, f6l, kBh[1], kBl[1], S);
    A32_SB();
    A32_LD_SCR(1, f9h, f9l);
    dA = mfma32x3(f7h, f7l, vBh[1], vBl[1], dA);
    A32_SB();
    // dQ^T[feature position][query] = K^T[.][key] . dS^T[key][query]  (B by transposed reads of the scratch image)
    f32x16 dQp = mfma32x3(kTh[0], kTl[0], f8h, f8l, zero16());
    dQp = mfma32x3(kTh[1], kTl[1], f9h, f9l, dQp);
    A32_SWEEP(4 + 4 * s);
    if (s > 0) {   // the accumulator's previous addition (wave w + 1, its step s - 1) must be in place
      const volatile int* f = sflag + (wave + 1 < NT ? wave + 1 : 0);
      while (*f < s) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }
    char* acc = DQACC + t * P_TILE + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4* a4 = reinterpret_cast<f32x4*>(acc + j * 1024);
      *a4 = *a4 + (f32x4){dQp[4 * j], dQp[4 * j + 1], dQp[4 * j + 2], dQp[4 * j + 3]};
    }
    asm volatile("" ::: "memory");
    *reinterpret_cast<volatile int*>(sflag + wave) = s + 1;   // LDS operations of a wave complete in order: the sums are in place
    A32_SWEEP(5 + 4 * s);
  }
  lds_barrier();   // every accumulator is complete

  // results for the wave's tokens: registers 8s .. 8s+7 = features 16s + 8h + (0..7)
  if (live) {
    f32x16 gq;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ownC + j * 1024 + lane * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) gq[4 * j + e] = a4[e] * p.qscale;
    }
    const f32x16 gk = dK * LN2;   // Q~ carries log2(e)
    float* dst = p.dqkv + (row0 + tok) * (3 * Hh) + 32 * head + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int o = 8 * s + 4 * hf;
        *reinterpret_cast<f32x4*>(dst + 16 * s + 4 * hf) = (f32x4){gq[o], gq[o + 1], gq[o + 2], gq[o + 3]};
        *reinterpret_cast<f32x4*>(dst + Hh + 16 * s + 4 * hf) = (f32x4){gk[o], gk[o + 1], gk[o + 2], gk[o + 3]};
        *reinterpret_cast<f32x4*>(dst + 2 * Hh + 16 * s + 4 * hf) = (f32x4){dV[o], dV[o + 1], dV[o + 2], dV[o + 3]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The attention core with the QUERIES restricted to the masked-LM slots (the last encoder layer of a train step or of an evaluation
// forward: only the rows the head gathers are read downstream, B4R_FLAG_HEAD_ROWS_ONLY).  Keys and values are all L tokens of the
// sequence, queries the P <= 64 slot positions: one or two COMPACT query tiles (query j of a sequence = its slot j, token
// clamp(position[b, j])), outputs in compact [B * P, .] tensors.  Forks of the two kernels above:
//   forward  -- every wave still writes the K / V images of its 32 tokens; waves 0 .. NQ - 1 then each sweep one compact query tile;
//   backward -- key owners as above, but a wave walks the NQ compact query tiles itself (no rotation, no step barrier) and keeps its dQ
//               partials in registers; they are summed over the key owners in wave order afterwards (ordered: bitwise reproducible).
// Dropout decisions are those of the (token query, key) pairs (the same counter hash index); the decision words live in a compact
// buffer of their own, [B][head][key tile][compact query tile][32].  Padded slots repeat position 0 with zero gradient; their dq rows
// are not written (a labelled slot may share that token).
// ---------------------------------------------------------------------------------------------------------------------------
struct A32SlotFwdP {
  const float* qkv; const int64_t* mask; const int64_t* pos;
  float* ctx_c; float* lse_c; uint32_t* bits_c;
  int B, L, NT, heads, P, NQ;
  DropArgs drop_p;
};

// forward: every wave owns the 32 KEYS of its tokens for both compact query tiles (S^T = K_w . Q~^T, a local softmax over its 32 keys,
// O^T partial = V_w^T . Pd^T); waves 0 .. NQ - 1 then merge the NT partials of their query tile (running maximum, rescaled sums) --
// the seven key tiles of a query are swept in parallel instead of one after the other by the query's wave.
// LDS (bytes): [K images NT x 4 KB | V images NT x 4 KB | Q~ images NQ x 4 KB | mask adders NT x 32 | (max, sum) NT x NQ x 32 x 2];
// a wave's O^T partials of query tile 0 / 1 go over its own (then dead) K / V image.
__host__ __device__ constexpr int slotq_fwd_lds(int NT, int NQ) { return (2 * NT + NQ) * P_TILE + (NT * 32 + NT * NQ * 64) * 4; }

template <int NQT, bool DROP>
__global__ __launch_bounds__(512, 2) void attn32_slotq_fwd_kernel(A32SlotFwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int NT = p.NT, L = p.L, NQ = p.NQ;
  char* const kimg = smem32;
  char* const vimg = smem32 + NT * P_TILE;
  char* const QIMG = vimg + NT * P_TILE;
  float* const sAdd = reinterpret_cast<float*>(QIMG + NQ * P_TILE);
  float* const sMS = sAdd + NT * 32;                   // [key tile][query tile][32 queries][max, sum]
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  const int Hh = 32 * p.heads;
  const int64_t row0 = (int64_t)b * L;
  const int64_t bh = (int64_t)b * p.heads + head;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  A32_LANE_CONSTS();
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  const float* const src = p.qkv + (row0 + tokc) * (3 * Hh) + 32 * head;
  f32x16 kacc = load_acc_layout(src + Hh, h), vacc = load_acc_layout(src + 2 * Hh, h);
  // the wave's compact queries (waves 0 .. NQ - 1)
  const bool qwave = wave < NQ;
  const int j = 32 * wave + r, jc = min(j, p.P - 1);
  const bool livej = qwave && j < p.P;
  f32x16 qa = zero16();
  if (qwave) {
    const int64_t pq = p.pos[(int64_t)b * p.P + jc];
    const int posq = (int)(pq < 0 ? 0 : (pq >= L ? L - 1 : pq));
    qa = load_acc_layout(p.qkv + (row0 + posq) * (3 * Hh) + 32 * head, h);
  }
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
  if ((int)threadIdx.x < NT * 32)
    sAdd[threadIdx.x] = (int)threadIdx.x < L ? (((1.0f - (float)mval) * -1e9f) - amax) * LOG2E : -INFINITY;
  qa = qa * (amax != 0.0f ? 0.0f : LOG2E);   // every key masked: Keras' -1e9 absorbs the scores -> a uniform softmax over the L keys
  if (qwave) acc_to_rows(QIMG + wave * P_TILE, lk, qa);
  if (!live) { kacc = zero16(); vacc = zero16(); }
  char* const kown = kimg + wave * P_TILE;
  char* const vown = vimg + wave * P_TILE;
  acc_to_rows(kown, lk, kacc);
  acc_to_rows(vown, lk, vacc);
  lds_barrier();

  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  const int slot = (r & 24) | ((r & 3) << 1) | ((r >> 2) & 1);
  bf16x8 kAh[2], kAl[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) { kAh[ks] = row_at(kown + lk.rowc[ks]); kAl[ks] = row_at(kown + P_IMG + lk.rowc[ks]); }
  const f32x16 add = rows_of(sAdd + 32 * wave, h);
  f32x16 O[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    O[t] = zero16();
    if (t >= NQ) continue;   // (wave-uniform)
    // the token of this lane's query in tile t (for the decisions' hash index)
    const int jt = min(32 * t + r, p.P - 1);
    const int64_t pq = p.pos[(int64_t)b * p.P + jt];
    const int post = (int)(pq < 0 ? 0 : (pq >= L ? L - 1 : pq));
    f32x16 S = add;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* qi = QIMG + t * P_TILE + lk.rowc[ks];
      S = mfma32x3(kAh[ks], kAl[ks], row_at(qi), row_at(qi + P_IMG), S);   // S^T[key][query] = K_w . Q~_t^T
    }
    float m = S[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) m = fmaxf(m, S[e]);
    m = fmaxf(m, other_half(m, h));
    const float mref = m == -INFINITY ? 0.0f : m;    // a tile of pad keys only: every term is 0
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { S[e] = __builtin_amdgcn_exp2f(S[e] - mref); sum += S[e]; }
    sum += other_half(sum, h);
    if (h == 0) { float* ms = sMS + ((wave * NQ + t) * 32 + r) * 2; ms[0] = m; ms[1] = sum; }
    if (DROP) {
      uint32_t word = 0;
      const uint64_t dbase = ((uint64_t)bh * L + (uint64_t)post) * (uint64_t)B4R_ATTN_PITCH;
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {
        const B4rKeep4 k4 = b4r_keep4p(dcp, dbase + (uint64_t)(32 * wave + 8 * gp + 4 * h));
#pragma unroll
        for (int e = 0; e < 4; ++e) S[4 * gp + e] = k4.k[e] ? S[4 * gp + e] : 0.f;
        word |= k4.bits() << (8 * gp + 4 * h);
      }
      word |= other_half_u(word, h);
      if (h == 0) p.bits_c[((bh * NT + wave) * NQ + t) * 32 + slot] = word;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 ph, pl;
      acc_frag(S, s2, ph, pl);
      O[t] = mfma32x3(tr_pair(vown + lk.trp[s2][0], vown + lk.trp[s2][1]), tr_pair(vown + P_IMG + lk.trp[s2][0], vown + P_IMG + lk.trp[s2][1]),
                      ph, pl, O[t]);
    }
  }
  // the partials over the wave's own images (every read of them above was the wave's own and has been consumed by a product)
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    if (t >= NQ) continue;
    char* dst = (t == 0 ? kown : vown) + lane * 16;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      *reinterpret_cast<f32x4*>(dst + jj * 1024) = (f32x4){O[t][4 * jj], O[t][4 * jj + 1], O[t][4 * jj + 2], O[t][4 * jj + 3]};
  }
  lds_barrier();
  if (!qwave) return;
  // merge of query tile t = wave: M = max_w m_w, sum = sum_w 2^(m_w - M) sum_w, O = sum_w 2^(m_w - M) O_w
  float M = -INFINITY;
  for (int w = 0; w < NT; ++w) M = fmaxf(M, sMS[((w * NQ + wave) * 32 + r) * 2]);
  float tot = 0.f;
  f32x16 Om = zero16();
  for (int w = 0; w < NT; ++w) {
    const float* ms = sMS + ((w * NQ + wave) * 32 + r) * 2;
    const float a = ms[0] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ms[0] - M);
    tot = fmaf(a, ms[1], tot);
    const char* srcp = (wave == 0 ? kimg : vimg) + w * P_TILE + lane * 16;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const f32x4 o4 = *reinterpret_cast<const f32x4*>(srcp + jj * 1024);
#pragma unroll
      for (int e = 0; e < 4; ++e) Om[4 * jj + e] = fmaf(a, o4[e], Om[4 * jj + e]);
    }
  }
  const float inv = (DROP ? dcp.scale : 1.0f) / tot;
  if (h == 0 && livej) p.lse_c[bh * p.P + j] = M * LN2 + __logf(tot);
  // O's registers 8s .. 8s+7 are context columns 16s + 8h + (0..7) of the head (V image in swapped column order, transposed reads)
  if (livej) {
    float* dst = p.ctx_c + ((int64_t)b * p.P + j) * Hh + 32 * head + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<f32x4*>(dst + 16 * s) = (f32x4){Om[8 * s] * inv, Om[8 * s + 1] * inv, Om[8 * s + 2] * inv, Om[8 * s + 3] * inv};
      *reinterpret_cast<f32x4*>(dst + 16 * s + 4) = (f32x4){Om[8 * s + 4] * inv, Om[8 * s + 5] * inv, Om[8 * s + 6] * inv, Om[8 * s + 7] * inv};
    }
  }
}

struct A32SlotBwdP {
  const float* qkv; const float* dctx_c; const float* ctx_c; const float* lse_c; const uint32_t* bits_c; const int64_t* mask;
  const int64_t* pos; const int64_t* ids;
  float* dqkv;
  int B, L, NT, heads, P, NQ;
  float qscale;
  DropArgs drop_p;
};
// LDS (bytes): [Q~ images NQ x 4 KB | dO images NQ x 4 KB | per-wave scratch NT x 4 KB | mask adders NT x 32, -lse and D NQ x 32 each]
__host__ __device__ constexpr int slotq_bwd_lds(int NT, int NQ) { return (2 * NQ + NT) * P_TILE + (NT * 32 + 2 * NQ * 32) * 4; }

template <int NQT, bool DROP>
__global__ __launch_bounds__(512, 2) void attn32_slotq_bwd_kernel(A32SlotBwdP p) {
  extern __shared__ __attribute__((aligned(16))) char smem32[];
  const int NT = p.NT, L = p.L, NQ = p.NQ;
  char* const QIMG = smem32;
  char* const DOIMG = QIMG + NQ * P_TILE;
  char* const SCRALL = DOIMG + NQ * P_TILE;
  float* const sAdd = reinterpret_cast<float*>(SCRALL + NT * P_TILE);
  float* const sCS = sAdd + NT * 32;
  float* const sD = sCS + NQ * 32;
  const int b = blockIdx.x / p.heads, head = blockIdx.x % p.heads;
  const int Hh = 32 * p.heads;
  const int64_t row0 = (int64_t)b * L;
  const int64_t bh = (int64_t)b * p.heads + head;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const scr = SCRALL + wave * P_TILE;
  A32_LANE_CONSTS();
  const int64_t mval = (int)threadIdx.x < L ? p.mask[row0 + threadIdx.x] : 0;
  const int any_key = mval != 0 ? 1 : 0;
  // key side: k, v of the wave's tokens; the q part of their dqkv rows is zero unless a labelled slot writes it below
  const float* const src = p.qkv + (row0 + tokc) * (3 * Hh) + 32 * head;
  f32x16 kT = load_acc_layout(src + Hh, h), vT = load_acc_layout(src + 2 * Hh, h);
  if (live) {
    float* dz = p.dqkv + (row0 + tok) * (3 * Hh) + 32 * head + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<f32x4*>(dz + 16 * s) = (f32x4){0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(dz + 16 * s + 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // query side (waves 0 .. NQ - 1): the compact rows
  const bool qwave = wave < NQ;
  const int j = 32 * wave + r, jc = min(j, p.P - 1);
  const bool livej = qwave && j < p.P;
  int posq = 0;
  if (qwave) {
    const int64_t m = (int64_t)b * p.P + jc;
    const int64_t pq = p.pos[m];
    posq = (int)(pq < 0 ? 0 : (pq >= L ? L - 1 : pq));
    f32x16 qT = load_acc_layout(p.qkv + (row0 + posq) * (3 * Hh) + 32 * head, h);
    const f32x16 dcT = load_acc_layout(p.dctx_c + m * Hh + 32 * head, h);
    const f32x16 cxT = load_acc_layout(p.ctx_c + m * Hh + 32 * head, h);
    const float lse_q = livej ? p.lse_c[bh * p.P + j] : INFINITY;
    qT = qT * LOG2E;
    float d = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) d = fmaf(dcT[e], cxT[e], d);
    d += other_half(d, h);
    if (h == 0) { sD[j] = d; sCS[j] = -lse_q * LOG2E; }
    acc_to_rows(QIMG + wave * P_TILE, lk, qT);
    acc_to_rows(DOIMG + wave * P_TILE, lk, dcT);
  }
  const float amax = __syncthreads_or(any_key) ? 0.0f : -1e9f;
  const bool dead = amax != 0.0f;
  if ((int)threadIdx.x < NT * 32)
    sAdd[threadIdx.x] = (int)threadIdx.x < L ? (((1.0f - (float)mval) * -1e9f) - amax) * LOG2E : -INFINITY;
  const DropCtx dcp = b4r_drop_ctx(p.drop_p);
  const float pscale = DROP ? dcp.scale : 1.0f;
  if (!live) { kT = zero16(); vT = zero16(); }
  bf16x8 kBh[2], kBl[2], vBh[2], vBl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) { acc_frag(kT, s, kBh[s], kBl[s]); acc_frag(vT, s, vBh[s], vBl[s]); }
  acc_to_rows(scr, lk, kT);
  lds_barrier();
  bf16x8 kTh[2], kTl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    kTh[s] = tr_pair(scr + lk.trn[s][0], scr + lk.trn[s][1]);
    kTl[s] = tr_pair(scr + P_IMG + lk.trn[s][0], scr + P_IMG + lk.trn[s][1]);
  }
  lds_barrier();   // query images, D, -lse and the mask adders in place; the K^T reads have landed
  const float adk = sAdd[tok];
  if (dead) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { kBh[ks][e] = (__bf16)0.f; kBl[ks][e] = (__bf16)0.f; }
  }
  typedef const __attribute__((address_space(4))) uint64_t* kmask_ptr;
  f32x16 dK = zero16(), dV = zero16(), dQp[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    dQp[t] = zero16();
    if (t >= NQ) continue;   // (wave-uniform)
    const char* qimg = QIMG + t * P_TILE;
    const char* dimg = DOIMG + t * P_TILE;
    f32x16 S = rows_of(sCS + 32 * t, h), dA = zero16();
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] += adk;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah, al;
      A32_LD_ROW(qimg, ks, ah, al);
      S = mfma32x3(ah, al, kBh[ks], kBl[ks], S);
      A32_LD_ROW(dimg, ks, ah, al);
      dA = mfma32x3(ah, al, vBh[ks], vBl[ks], dA);
    }
    const f32x16 Dq = rows_of(sD + 32 * t, h);
    uint64_t km[16];
    if (DROP) {
      kmask_ptr mp = (kmask_ptr)(reinterpret_cast<const uint64_t*>(p.bits_c) + ((bh * NT + wave) * NQ + t) * 16);
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) km[tt] = mp[tt];
    }
    bf16x8 pdh[2], pdl[2], dsh[2], dsl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      f32x8 pd, ds;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int tt = 8 * s2 + jj;
        const float pr = __builtin_amdgcn_exp2f(S[tt]);
        if (DROP) {
          const float kf = __builtin_amdgcn_inverse_ballot_w64(km[tt]) ? pscale : 0.f;
          pd[jj] = pr * kf;
          ds[jj] = pr * fmaf(dA[tt], kf, -Dq[tt]);
        } else {
          pd[jj] = pr;
          ds[jj] = pr * (dA[tt] - Dq[tt]);
        }
      }
      split8(pd, pdh[s2], pdl[s2]);
      split8(ds, dsh[s2], dsl[s2]);
      const s16x8 hv = __builtin_bit_cast(s16x8, dsh[s2]), lv = __builtin_bit_cast(s16x8, dsl[s2]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        char* w8 = scr + p_chunk(r, 2 * s2 + a) + 8 * h;
        *reinterpret_cast<s16x4*>(w8) = a ? __builtin_shufflevector(hv, hv, 4, 5, 6, 7) : __builtin_shufflevector(hv, hv, 0, 1, 2, 3);
        *reinterpret_cast<s16x4*>(w8 + P_IMG) = a ? __builtin_shufflevector(lv, lv, 4, 5, 6, 7) : __builtin_shufflevector(lv, lv, 0, 1, 2, 3);
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 fh, fl;
      A32_LD_TR(dimg, s2, fh, fl);
      dV = mfma32x3(fh, fl, pdh[s2], pdl[s2], dV);   // dV^T[feature][key] += dO^T[feature][query] . Pd[query][key]
      A32_LD_TR(qimg, s2, fh, fl);
      dK = mfma32x3(fh, fl, dsh[s2], dsl[s2], dK);   // dK^T += Q~^T . dS
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fh, fl;
      A32_LD_SCR(ks, fh, fl);
      dQp[t] = mfma32x3(kTh[ks], kTl[ks], fh, fl, dQp[t]);   // dQ^T[feature position][query] = K^T . dS^T
    }
  }
  // results of the wave's tokens as keys: registers 8s .. 8s+7 = features 16s + 8h + (0..7)
  if (live) {
    const f32x16 gk = dK * LN2;   // Q~ carries log2(e)
    float* dst = p.dqkv + (row0 + tok) * (3 * Hh) + 32 * head + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int o = 8 * s + 4 * hf;
        *reinterpret_cast<f32x4*>(dst + Hh + 16 * s + 4 * hf) = (f32x4){gk[o], gk[o + 1], gk[o + 2], gk[o + 3]};
        *reinterpret_cast<f32x4*>(dst + 2 * Hh + 16 * s + 4 * hf) = (f32x4){dV[o], dV[o + 1], dV[o + 2], dV[o + 3]};
      }
  }
  // dQ of compact tile t = the key owners' partials in wave order (through the scratch tiles, one compact tile at a time); the zero
  // fill of the q parts above is complete before the first labelled row is written (vmcnt(0) in front of the barrier)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < NQT; ++t) {
    if (t >= NQ) continue;
    lds_barrier();   // (t == 0: the sweep's last scratch reads; t > 0: the previous tile's sums have been read)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      *reinterpret_cast<f32x4*>(scr + jj * 1024 + lane * 16) = (f32x4){dQp[t][4 * jj], dQp[t][4 * jj + 1], dQp[t][4 * jj + 2], dQp[t][4 * jj + 3]};
    lds_barrier();
    if (wave == t) {
      f32x16 gq = zero16();
      for (int w = 0; w < NT; ++w)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(SCRALL + w * P_TILE + jj * 1024 + lane * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) gq[4 * jj + e] += a4[e];
        }
      if (livej && p.ids[(int64_t)b * p.P + j] != 0) {
        float* dst = p.dqkv + (row0 + posq) * (3 * Hh) + 32 * head + 8 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int o = 8 * s + 4 * hf;
            *reinterpret_cast<f32x4*>(dst + 16 * s + 4 * hf) =
                (f32x4){gq[o] * p.qscale, gq[o + 1] * p.qscale, gq[o + 2] * p.qscale, gq[o + 3] * p.qscale};
          }
      }
    }
  }
}

bool al16(const void* q) { return q == nullptr || b4r_aligned16(q); }

}  // namespace

int32_t b4r_attn32_supported(int32_t hidden_size, int32_t num_heads, int32_t L) {
  return (hidden_size == HID && num_heads == 2 && L > 0 && L <= 224 && b4r_get_gemm_mode() == B4R_GEMM_BF16X3) ? 1 : 0;
}
// Sequences of at most two 32-token tiles leave five of the seven waves of a workgroup without a query / key tile: there round 2's
// 16-token-tile kernels (one wave per 16 tokens) are faster -- Steam, L = 50: 0.408 against 0.440 ms per train step.  The 32-token-tile
// kernels are PREFERRED from this length on (they still serve shorter sequences when a descriptor asks for what only they can do).
static int g_attn32_min_len = getenv("B4R_ATTN32_MIN_L") ? atoi(getenv("B4R_ATTN32_MIN_L")) : 65;
extern "C" int32_t b4r_attn32_set_min_len(int32_t L) {
  const int old = g_attn32_min_len;
  if (L >= 0) g_attn32_min_len = L;
  return old;
}
int32_t b4r_attn32_preferred(int32_t hidden_size, int32_t num_heads, int32_t L) {
  return (b4r_attn32_supported(hidden_size, num_heads, L) && L >= g_attn32_min_len) ? 1 : 0;
}

// ---- the attention core for any number of 32-wide heads (b4r_attn_fwd / b4r_attn_bwd in the bf16x3 mode) ----------------------
int64_t b4r_attn_rx_keep_words(int B, int L, int heads);
bool b4r_attn32_core_preferred(int L) {
  static const bool on = !(getenv("B4R_ATTN32_CORE") && atoi(getenv("B4R_ATTN32_CORE")) == 0);
  return on && L > 0 && L <= 224 && L >= g_attn32_min_len && b4r_get_gemm_mode() == B4R_GEMM_BF16X3;
}
static int g_attn32_core_fwd = getenv("B4R_ATTN32_CORE_FWD") ? atoi(getenv("B4R_ATTN32_CORE_FWD")) : 0;
bool b4r_attn32_core_fwd_wanted() { return g_attn32_core_fwd != 0; }
extern "C" int32_t b4r_attn32_set_core_fwd(int32_t on) {
  const int old = g_attn32_core_fwd;
  if (on >= 0) g_attn32_core_fwd = on;
  return old;
}
int b4r_attn32_core_fwd_launch(const float* qkv, const int64_t* mask, int B, int L, int heads, float* ctx, float* lse,
                               const DropArgs& drop, uint32_t* keep_bits, hipStream_t stream) {
  A32CoreFwdP p{};
  p.qkv = qkv; p.mask = mask; p.ctx = ctx; p.lse = lse;
  p.bits = keep_bits ? keep_bits + b4r_attn_rx_keep_words(B, L, heads) : nullptr;   // behind round 1's layout
  p.B = B; p.L = L; p.NT = b4r_cdiv(L, 32); p.heads = heads;
  p.drop_p = drop;
  const bool dropping = drop.rng != nullptr && drop.thr != 0;
  B4R_CHECK_ARG(!dropping || keep_bits, B4R_E_BADARG, "b4r_attn_fwd: attention dropout needs keep_bits");
  const size_t sh = (size_t)core_fwd_lds(p.NT);
  const dim3 grid((unsigned)(B * heads)), block((unsigned)(64 * p.NT));
  int rc;
#define A32_CORE_FWD_CASE(N_, D_)                                                                       \
  {                                                                                                     \
    rc = b4r_raise_lds((const void*)attn32_core_fwd_kernel<N_, D_>, sh, "b4r_attn_fwd");                \
    if (rc) return rc;                                                                                  \
    hipLaunchKernelGGL((attn32_core_fwd_kernel<N_, D_>), grid, block, sh, stream, p);                   \
  }
  if (p.NT <= 2) { if (dropping) A32_CORE_FWD_CASE(2, true) else A32_CORE_FWD_CASE(2, false) }
  else if (p.NT <= 4) { if (dropping) A32_CORE_FWD_CASE(4, true) else A32_CORE_FWD_CASE(4, false) }
  else { if (dropping) A32_CORE_FWD_CASE(7, true) else A32_CORE_FWD_CASE(7, false) }
#undef A32_CORE_FWD_CASE
  B4R_CHECK_LAUNCH("b4r_attn_fwd (32-token tiles)");
  return B4R_OK;
}
int b4r_attn32_core_bwd_launch(const float* qkv, const int64_t* mask, const float* ctx, const float* lse, const float* dctx, int B,
                               int L, int heads, float qscale, float* dqkv, const DropArgs& drop, const uint32_t* keep_bits,
                               hipStream_t stream) {
  A32CoreBwdP p{};
  p.qkv = qkv; p.mask = mask; p.ctx = ctx; p.lse = lse; p.dctx = dctx; p.dqkv = dqkv;
  p.bits = keep_bits ? keep_bits + b4r_attn_rx_keep_words(B, L, heads) : nullptr;
  p.B = B; p.L = L; p.NT = b4r_cdiv(L, 32); p.heads = heads; p.qscale = qscale;
  p.drop_p = drop;
  const bool dropping = drop.rng != nullptr && drop.thr != 0;
  B4R_CHECK_ARG(!dropping || keep_bits, B4R_E_BADARG, "b4r_attn_bwd: attention dropout needs the forward's keep_bits");
  const size_t sh = (size_t)core_bwd_lds(p.NT);
  const dim3 grid((unsigned)(B * heads)), block((unsigned)(64 * p.NT));
  int rc;
  if (dropping) {
    rc = b4r_raise_lds((const void*)attn32_core_bwd_kernel<true>, sh, "b4r_attn_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL((attn32_core_bwd_kernel<true>), grid, block, sh, stream, p);
  } else {
    rc = b4r_raise_lds((const void*)attn32_core_bwd_kernel<false>, sh, "b4r_attn_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL((attn32_core_bwd_kernel<false>), grid, block, sh, stream, p);
  }
  B4R_CHECK_LAUNCH("b4r_attn_bwd (32-token tiles)");
  return B4R_OK;
}

// attention-dropout decisions of one layer: [B][head][key tile][query tile][16 register pairs] x 2 uint32
int64_t b4r_attn32_keep_words(int32_t B, int32_t L, int32_t heads) {
  const int64_t NT = b4r_cdiv(L, 32);
  return (int64_t)B * heads * NT * NT * 32;
}

// ---- the attention core on the masked-LM slots' queries only (b4r_model.hip: the last layer under B4R_FLAG_HEAD_ROWS_ONLY) -------------
bool b4r_attn32_slotq_supported(int L, int P) {
  static const bool on = !(getenv("B4R_ATTN_SLOTQ") && atoi(getenv("B4R_ATTN_SLOTQ")) == 0);
  return on && L > 64 && L <= 224 && P > 0 && P <= 64 && 2 * P <= L && b4r_get_gemm_mode() == B4R_GEMM_BF16X3;
}
int64_t b4r_attn32_slotq_keep_words(int B, int L, int heads, int P) {
  return (int64_t)B * heads * b4r_cdiv(L, 32) * b4r_cdiv(P, 32) * 32;
}
int b4r_attn32_slotq_fwd_launch(const float* qkv, const int64_t* mask, const int64_t* pos, int B, int L, int heads, int P, float* ctx_c,
                                float* lse_c, const DropArgs& drop, uint32_t* bits_c, hipStream_t stream) {
  A32SlotFwdP p{};
  p.qkv = qkv; p.mask = mask; p.pos = pos; p.ctx_c = ctx_c; p.lse_c = lse_c; p.bits_c = bits_c;
  p.B = B; p.L = L; p.NT = b4r_cdiv(L, 32); p.heads = heads; p.P = P; p.NQ = b4r_cdiv(P, 32);
  p.drop_p = drop;
  const bool dropping = drop.rng != nullptr && drop.thr != 0;
  B4R_CHECK_ARG(!dropping || bits_c, B4R_E_BADARG, "attention on the slots' queries: dropout needs the decision buffer");
  const size_t sh = (size_t)slotq_fwd_lds(p.NT, p.NQ);
  const dim3 grid((unsigned)(B * heads)), block((unsigned)(64 * p.NT));
  int rc;
#define A32_SLOT_FWD_CASE(Q_, D_)                                                                        \
  {                                                                                                      \
    rc = b4r_raise_lds((const void*)attn32_slotq_fwd_kernel<Q_, D_>, sh, "attention on the slots' queries"); \
    if (rc) return rc;                                                                                   \
    hipLaunchKernelGGL((attn32_slotq_fwd_kernel<Q_, D_>), grid, block, sh, stream, p);                   \
  }
  if (p.NQ <= 1) { if (dropping) A32_SLOT_FWD_CASE(1, true) else A32_SLOT_FWD_CASE(1, false) }
  else { if (dropping) A32_SLOT_FWD_CASE(2, true) else A32_SLOT_FWD_CASE(2, false) }
#undef A32_SLOT_FWD_CASE
  B4R_CHECK_LAUNCH("attention core forward, queries = the head's slots");
  return B4R_OK;
}
int b4r_attn32_slotq_bwd_launch(const float* qkv, const int64_t* mask, const int64_t* pos, const int64_t* ids, const float* ctx_c,
                                const float* lse_c, const float* dctx_c, int B, int L, int heads, int P, float qscale, float* dqkv,
                                const DropArgs& drop, const uint32_t* bits_c, hipStream_t stream) {
  A32SlotBwdP p{};
  p.qkv = qkv; p.mask = mask; p.pos = pos; p.ids = ids; p.ctx_c = ctx_c; p.lse_c = lse_c; p.dctx_c = dctx_c; p.dqkv = dqkv; p.bits_c = bits_c;
  p.B = B; p.L = L; p.NT = b4r_cdiv(L, 32); p.heads = heads; p.P = P; p.NQ = b4r_cdiv(P, 32); p.qscale = qscale;
  p.drop_p = drop;
  const bool dropping = drop.rng != nullptr && drop.thr != 0;
  B4R_CHECK_ARG(!dropping || bits_c, B4R_E_BADARG, "attention on the slots' queries: dropout needs the forward's decisions");
  const size_t sh = (size_t)slotq_bwd_lds(p.NT, p.NQ);
  const dim3 grid((unsigned)(B * heads)), block((unsigned)(64 * p.NT));
  int rc;
#define A32_SLOT_BWD_CASE(Q_, D_)                                                                        \
  {                                                                                                      \
    rc = b4r_raise_lds((const void*)attn32_slotq_bwd_kernel<Q_, D_>, sh, "attention on the slots' queries"); \
    if (rc) return rc;                                                                                   \
    hipLaunchKernelGGL((attn32_slotq_bwd_kernel<Q_, D_>), grid, block, sh, stream, p);                   \
  }
  if (p.NQ <= 1) { if (dropping) A32_SLOT_BWD_CASE(1, true) else A32_SLOT_BWD_CASE(1, false) }
  else { if (dropping) A32_SLOT_BWD_CASE(2, true) else A32_SLOT_BWD_CASE(2, false) }
#undef A32_SLOT_BWD_CASE
  B4R_CHECK_LAUNCH("attention core backward, queries = the head's slots");
  return B4R_OK;
}

#ifdef A32_PROF
extern "C" int b4r_debug_a32_prof(long long* host_out) {   // 64 stamps of the last backward launch (after a device synchronisation)
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a32_prof), 64 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
extern "C" int b4r_debug_a32f_prof(long long* host_out) {   // stamps of the last forward launch
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a32f_prof), 32 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
extern "C" int b4r_debug_a32_sweep(long long* host_out) {   // [2 waves][64] stamps inside the sweep of head 0
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a32_sweep), 128 * sizeof(long long)) == hipSuccess ? 0 : -4;
}
#endif
int64_t b4r_attn_rx_keep_words(int B, int L, int heads);
int b4r_launch_slab_reduce_full(const float* slab, int S, int Mo, int No, float* out, int ldo, int accumulate,
                                const float* cslab, float* colsum, const float* caslab, float* colsum_a, hipStream_t stream);

int b4r_attn32_bwd(const b4r_attn_block_bwd_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_attn_block_bwd: null descriptor");
  B4R_CHECK_ARG(b4r_attn32_supported(d->H, d->heads, d->L), B4R_E_SHAPE,
                "b4r_attn_block_bwd: needs hidden size 64, 2 heads, L <= 224 and the bf16x3 mode (H=%d heads=%d L=%d)", d->H, d->heads,
                d->L);
  B4R_CHECK_ARG(d->B > 0 && d->x && d->dz1 && d->ctx && d->lse && d->input_mask && d->Wqkv && d->bqkv && d->Wo && d->prev_mean &&
                    d->prev_rstd && d->prev_gamma && d->dx_prev && d->dprev_gamma && d->scratch,
                B4R_E_BADARG, "b4r_attn_block_bwd: null argument");
  B4R_CHECK_ARG(d->dqkv || d->dWqkv, B4R_E_BADARG, "b4r_attn_block_bwd: needs dqkv, or dWqkv + dbqkv + dw_scratch");
  B4R_CHECK_ARG(!d->dWqkv || (d->dbqkv && d->dw_scratch), B4R_E_BADARG, "b4r_attn_block_bwd: dWqkv needs dbqkv and dw_scratch");
  const bool embed = d->emb_ids != nullptr;
  B4R_CHECK_ARG(embed ? (d->emb_table && d->emb_pos && d->emb_vocab > 0) : (d->prev_z != nullptr), B4R_E_BADARG,
                "b4r_attn_block_bwd: needs prev_z, or emb_ids + emb_table + emb_pos");
  B4R_CHECK_ARG(al16(d->x) && al16(d->dz1) && al16(d->ctx) && al16(d->Wqkv) && al16(d->Wo) && al16(d->prev_z) && al16(d->prev_gamma) &&
                    al16(d->emb_table) && al16(d->emb_pos) && al16(d->dqkv) && al16(d->dx_prev) && al16(d->keep_bits) && al16(d->dw_scratch),
                B4R_E_ALIGN, "b4r_attn_block_bwd: operands must be 16-byte aligned");
  A32BwdP p{};
  p.x = d->x; p.dz1 = d->dz1; p.ctx = d->ctx; p.lse = d->lse; p.mask = d->input_mask;
  p.bits = d->keep_bits ? d->keep_bits + b4r_attn_rx_keep_words(d->B, d->L, d->heads) : nullptr;   // behind round 1's layout
  p.Wqkv = d->Wqkv; p.bqkv = d->bqkv; p.Wo = d->Wo;
  p.zprev = d->prev_z; p.meanp = d->prev_mean; p.rstdp = d->prev_rstd; p.gprev = d->prev_gamma;
  p.ids = d->emb_ids; p.table = d->emb_table; p.pos = d->emb_pos; p.V = d->emb_vocab;
  p.dqkv = d->dqkv; p.da = d->dx_prev; p.ln_part = d->scratch;
  B4R_CHECK_ARG((d->dz1_slot_positions == nullptr) == (d->dz1_slot_ids == nullptr) && (d->dz1_slot_positions == nullptr || d->dz1_slots > 0),
                B4R_E_BADARG, "b4r_attn_block_bwd: dz1_slot_positions, dz1_slot_ids and dz1_slots go together");
  p.slot_pos = d->dz1_slot_positions; p.slot_ids = d->dz1_slot_ids; p.slots = d->dz1_slots;
  const bool fold = d->dWqkv != nullptr;
  if (fold) { p.dw_slab = d->dw_scratch; p.db_slab = d->dw_scratch + (int64_t)d->B * (HID * 3 * HID); }
  const bool fold_wo = d->dWo != nullptr;
  B4R_CHECK_ARG(!fold_wo || (fold && d->dbo), B4R_E_BADARG, "b4r_attn_block_bwd: dWo needs dbo and the dWqkv outputs");
  if (fold_wo) { p.dwo_slab = p.db_slab + (int64_t)d->B * (3 * HID); p.dbo_slab = p.dwo_slab + (int64_t)d->B * (HID * HID); }
  p.B = d->B; p.L = d->L; p.NT = b4r_cdiv(d->L, 32);
  p.qscale = 1.0f / sqrtf(32.0f);
  p.drop_p = b4r_make_drop(d->rng, d->probs_stream, d->probs_rate, d->rng != nullptr);
  p.drop_o = b4r_make_drop(d->rng, d->out_stream, d->out_rate, d->rng != nullptr);
  p.drop_e = b4r_make_drop(d->rng, d->emb_stream, d->emb_rate, d->rng != nullptr && embed);
  B4R_CHECK_ARG(!p.drop_p.rng || d->keep_bits, B4R_E_BADARG, "b4r_attn_block_bwd: attention dropout needs the forward's keep_bits");
  // the slots as the sweep's only queries (the kernel's CQ form): a row list of at most 64 slots on at least three token tiles
  static const bool cq_on = !(getenv("B4R_ATTN32_CQ") && atoi(getenv("B4R_ATTN32_CQ")) == 0);
  const bool cq = cq_on && p.slot_pos != nullptr && p.slots <= 64 && p.NT >= 3;
  const size_t sh = cq ? (size_t)bwd32_lds_cq(p.NT) : (size_t)bwd32_lds(p.NT);
  const dim3 grid((unsigned)d->B), block((unsigned)(64 * p.NT));
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const bool drop = p.drop_p.rng != nullptr && p.drop_p.thr != 0;
#define A32_BWD_CASE(E_, D_, C_)                                                                      \
  {                                                                                                   \
    rc = b4r_raise_lds((const void*)attn32_bwd_kernel<E_, D_, C_>, sh, "b4r_attn_block_bwd");         \
    if (rc) return rc;                                                                                \
    hipLaunchKernelGGL((attn32_bwd_kernel<E_, D_, C_>), grid, block, sh, s, p);                       \
  }
  if (cq) {
    if (embed) { if (drop) A32_BWD_CASE(true, true, true) else A32_BWD_CASE(true, false, true) }
    else { if (drop) A32_BWD_CASE(false, true, true) else A32_BWD_CASE(false, false, true) }
  } else {
    if (embed) { if (drop) A32_BWD_CASE(true, true, false) else A32_BWD_CASE(true, false, false) }
    else { if (drop) A32_BWD_CASE(false, true, false) else A32_BWD_CASE(false, false, false) }
  }
#undef A32_BWD_CASE
  B4R_CHECK_LAUNCH("b4r_attn_block_bwd");
  // gamma / beta gradients of the previous LayerNorm: ordered sum over the sequences (queued with the caller's reductions)
  rc = b4r_launch_slab_reduce_full(d->scratch, d->B, 1, 128, d->dprev_gamma, 128, 0, nullptr, nullptr, nullptr, nullptr, s);
  if (rc != B4R_OK || !fold) return rc;
  rc = b4r_launch_slab_reduce_full(p.dw_slab, d->B, HID, 3 * HID, d->dWqkv, 3 * HID, 0, p.db_slab, d->dbqkv, nullptr, nullptr, s);
  if (rc != B4R_OK || !fold_wo) return rc;
  return b4r_launch_slab_reduce_full(p.dwo_slab, d->B, HID, HID, d->dWo, HID, 0, p.dbo_slab, d->dbo, nullptr, nullptr, s);
}

int b4r_attn32_fwd(const b4r_attn_block_desc* d, b4r_stream_t stream) {
  B4R_CHECK_ARG(d != nullptr, B4R_E_BADARG, "b4r_attn_block_fwd: null descriptor");
  B4R_CHECK_ARG(b4r_attn32_supported(d->H, d->heads, d->L), B4R_E_SHAPE,
                "b4r_attn_block_fwd: needs hidden size 64, 2 heads, L <= 224 and the bf16x3 mode (H=%d heads=%d L=%d)", d->H, d->heads, d->L);
  const bool embed = d->emb_ids != nullptr;
  B4R_CHECK_ARG(d->B > 0 && (d->x || embed) && d->input_mask && d->Wqkv && d->bqkv && d->Wo && d->bo && d->ln_gamma && d->ln_beta &&
                    (d->x1 || (d->z1 && d->mean1 && d->rstd1)),
                B4R_E_BADARG, "b4r_attn_block_fwd: null argument (outputs: x1, or z1 + mean1 + rstd1)");
  B4R_CHECK_ARG(!embed || (d->emb_table && d->emb_pos && d->emb_gamma && d->emb_beta && d->emb_x && d->emb_vocab > 0 &&
                           al16(d->emb_table) && al16(d->emb_pos) && al16(d->emb_gamma) && al16(d->emb_beta) && al16(d->emb_x)),
                B4R_E_BADARG, "b4r_attn_block_fwd: the embedding mode needs emb_table, emb_pos, emb_gamma, emb_beta, emb_x (16-byte "
                "aligned) and emb_vocab");
  B4R_CHECK_ARG(al16(d->x) && al16(d->Wqkv) && al16(d->Wo) && al16(d->bo) && al16(d->ln_gamma) && al16(d->ln_beta) && al16(d->qkv) &&
                    al16(d->ctx) && al16(d->z1) && al16(d->x1) && al16(d->keep_bits),
                B4R_E_ALIGN, "b4r_attn_block_fwd: operands must be 16-byte aligned");
  A32FwdP p{};
  p.x = d->x; p.mask = d->input_mask; p.Wqkv = d->Wqkv; p.bqkv = d->bqkv; p.Wo = d->Wo; p.bo = d->bo;
  p.g1 = d->ln_gamma; p.be1 = d->ln_beta; p.qkv = d->qkv; p.ctx = d->ctx; p.lse = d->lse;
  p.bits = d->keep_bits ? d->keep_bits + b4r_attn_rx_keep_words(d->B, d->L, d->heads) : nullptr;   // behind round 1's layout
  p.z1 = d->z1; p.x1 = d->x1; p.mean1 = d->mean1; p.rstd1 = d->rstd1;
  p.B = d->B; p.L = d->L; p.NT = b4r_cdiv(d->L, 32);
  p.qscale = 1.0f / sqrtf(32.0f); p.eps = d->ln_eps;
  p.drop_p = b4r_make_drop(d->rng, d->probs_stream, d->probs_rate, d->rng != nullptr);
  p.drop_o = b4r_make_drop(d->rng, d->out_stream, d->out_rate, d->rng != nullptr);
  B4R_CHECK_ARG(!p.drop_p.rng || d->keep_bits, B4R_E_BADARG, "b4r_attn_block_fwd: attention dropout needs keep_bits");
  if (embed) {
    p.ids = d->emb_ids; p.table = d->emb_table; p.pos = d->emb_pos; p.g0 = d->emb_gamma; p.be0 = d->emb_beta; p.V = d->emb_vocab;
    p.x_out = d->emb_x; p.mean0 = d->emb_mean; p.rstd0 = d->emb_rstd; p.eps0 = d->emb_eps;
    p.drop_e = b4r_make_drop(d->rng, d->emb_stream, d->emb_rate, d->rng != nullptr);
  }
  const bool drop = p.drop_p.rng != nullptr && p.drop_p.thr != 0;
  // the slots as the only queries (the kernel's CQ form): a list of at most 64 rows on at least three token tiles, not the first layer
  // (B4R_ATTN32_CQ=0 switches the backward's form off: it then reads every token's ctx / lse, so the forward must write them all)
  static const bool cq_on = !(getenv("B4R_ATTN32_CQ_FWD") && atoi(getenv("B4R_ATTN32_CQ_FWD")) == 0) &&
                            !(getenv("B4R_ATTN32_CQ") && atoi(getenv("B4R_ATTN32_CQ")) == 0);
  B4R_CHECK_ARG(d->out_slot_positions == nullptr || d->out_slots > 0, B4R_E_BADARG, "b4r_attn_block_fwd: out_slot_positions needs out_slots");
  const bool cq = cq_on && d->out_slot_positions != nullptr && d->out_slots <= 64 && p.NT >= 3 && !embed && d->qkv == nullptr;
  if (cq) { p.slot_pos = d->out_slot_positions; p.slots = d->out_slots; }
  const size_t sh = cq ? (size_t)fwd32_lds_cq(p.NT) : (size_t)fwd32_lds(p.NT);
  const dim3 grid((unsigned)d->B), block((unsigned)(64 * p.NT));
  hipStream_t s = (hipStream_t)stream;
  int rc;
#define A32_FWD_CASE(N_, D_, C_)                                                                      \
  {                                                                                                   \
    rc = b4r_raise_lds((const void*)attn32_fwd_kernel<N_, D_, C_>, sh, "b4r_attn_block_fwd");         \
    if (rc) return rc;                                                                                \
    hipLaunchKernelGGL((attn32_fwd_kernel<N_, D_, C_>), grid, block, sh, s, p);                       \
  }
  if (cq) { if (drop) A32_FWD_CASE(7, true, true) else A32_FWD_CASE(7, false, true) }
  else if (p.NT <= 2) { if (drop) A32_FWD_CASE(2, true, false) else A32_FWD_CASE(2, false, false) }
  else if (p.NT <= 4) { if (drop) A32_FWD_CASE(4, true, false) else A32_FWD_CASE(4, false, false) }
  else { if (drop) A32_FWD_CASE(7, true, false) else A32_FWD_CASE(7, false, false) }
#undef A32_FWD_CASE
  B4R_CHECK_LAUNCH("b4r_attn_block_fwd");
  return B4R_OK;
}
