// Tile records of the 32 x 32-tile masked-LM head (b4r_head32.hip): fp16 hi / lo panel images of 32 rows + a side block.  In a header
// because the conversion can RIDE on another launch as extra workgroups (b4r_zero2: the transform rows' records with -lse and the
// labels are formed beside the gradient clear that opens the backward -- one launch boundary less per step).
#pragma once
#include "b4r_tile32.h"
#include "b4r_head_merge.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int H32_SIDE = 256;                      // bytes behind a tile's images: 32 floats (bias | -lse) + 32 ints (labels)
__host__ __device__ constexpr int h32_rec(int np) { return np * P_TILE + H32_SIDE; }   // bytes of one 32-row tile record

// x = hi + lo, hi = fp16(x), lo = fp16(x - hi): 22 significant bits while |x| stays in fp16's normal range (6e-5 .. 65504; below it
// lo keeps fewer bits, never fewer than bf16's split in total for |x| >= 1e-3)
__device__ __forceinline__ void h32_split_pair(float a, float b, uint32_t& hw, uint32_t& lw) {
  const f16x2 hh = __builtin_convertvector((b4r_f32x2){a, b}, f16x2);
  const b4r_f32x2 back = __builtin_convertvector(hh, b4r_f32x2);
  hw = __builtin_bit_cast(uint32_t, hh);
  lw = __builtin_bit_cast(uint32_t, __builtin_convertvector((b4r_f32x2){a - back[0], b - back[1]}, f16x2));
}
__device__ __forceinline__ void h32_split4(const f32x4 x, f16x4& hi, f16x4& lo) {
  b4r_u32x2 hw, lw;
#pragma unroll
  for (int j = 0; j < 2; ++j) { uint32_t a, b; h32_split_pair(x[2 * j], x[2 * j + 1], a, b); hw[j] = a; lw[j] = b; }
  hi = __builtin_bit_cast(f16x4, hw); lo = __builtin_bit_cast(f16x4, lw);
}
__device__ __forceinline__ void h32_split8(const f32x8 x, f16x8& hi, f16x8& lo) {
  b4r_u32x4 hw, lw;
#pragma unroll
  for (int j = 0; j < 4; ++j) { uint32_t a, b; h32_split_pair(x[2 * j], x[2 * j + 1], a, b); hw[j] = a; lw[j] = b; }
  hi = __builtin_bit_cast(f16x8, hw); lo = __builtin_bit_cast(f16x8, lw);
}
// ---------------------------------------------------------------------------------------------------------------------------
// pack: rows of a [R, 32 NP] fp32 matrix -> one record per 32-row tile: NP panel tiles (hi image | lo image, b4r_tile32.h layout,
// rows beyond R zero) + the side block.  One workgroup per tile.
//   mode 0 (E): side[j] = bias[row] * log2(e), -inf beyond R
//   mode 1 (T): side[j] = -lse[row] * log2(e) (-inf: no label / beyond R), side[32 + j] = label (-1: none); lse / label either given or
//               (cpart != NULL) formed from the forward's per-slice (max, sum) pairs exactly as head_merge_row forms them
// ---------------------------------------------------------------------------------------------------------------------------
struct H32PackP {
  const float* src; int R; char* dst; int mode; int np;
  const float* bias;
  const float* lse; const int32_t* ylab;
  const float* cpart; int cslices; const int64_t* y; int V;
};

// one tile, by the 256 threads of a workgroup
template <int NP>
__device__ __forceinline__ void h32_pack_tile(const H32PackP& p, int tile) {
  constexpr int H = 32 * NP, REC = h32_rec(NP);
  char* rec = p.dst + (int64_t)tile * REC;
  f32x4 v[NP];
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int f = threadIdx.x + 256 * it, row = f / (8 * NP), q = f % (8 * NP);
    const int gr = min(32 * tile + row, p.R - 1);
    v[it] = *reinterpret_cast<const f32x4*>(p.src + (int64_t)gr * H + 4 * q);
  }
#pragma unroll
  for (int it = 0; it < NP; ++it) {
    const int f = threadIdx.x + 256 * it, row = f / (8 * NP), q = f % (8 * NP);
    const f32x4 x = (32 * tile + row < p.R) ? v[it] : (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x4 hh, ll;
    h32_split4(x, hh, ll);
    char* d8 = rec + (q >> 3) * P_TILE + p_chunk(row, (q & 7) >> 1) + 8 * (q & 1);
    *reinterpret_cast<f16x4*>(d8) = hh;
    *reinterpret_cast<f16x4*>(d8 + P_IMG) = ll;
  }
  if (threadIdx.x < 32) {
    const int gr = 32 * tile + threadIdx.x;
    const bool in = gr < p.R;
    float* side = reinterpret_cast<float*>(rec + NP * P_TILE);
    if (p.mode == 0) {
      side[threadIdx.x] = in ? p.bias[gr] * LOG2E : -INFINITY;
      side[32 + threadIdx.x] = 0.f;
    } else {
      float lz = INFINITY;
      int yz = -1;
      if (p.cpart != nullptr) {
        RowPart rp;
        row_part_fetch(rp, p.cpart, p.cslices, p.R, p.y, min(gr, p.R - 1));
        row_part_finish(rp, p.V, lz, yz);
      } else {
        lz = p.lse[min(gr, p.R - 1)];
        yz = p.ylab[min(gr, p.R - 1)];
      }
      side[threadIdx.x] = in ? -(lz * LOG2E) : -INFINITY;
      reinterpret_cast<int*>(side)[32 + threadIdx.x] = in ? yz : -1;
    }
  }
}

__device__ __forceinline__ void h32_pack_tile_any(const H32PackP& p, int tile) {
  if (p.np == 2) h32_pack_tile<2>(p, tile);
  else if (p.np == 4) h32_pack_tile<4>(p, tile);
  else h32_pack_tile<8>(p, tile);
}

}  // namespace
