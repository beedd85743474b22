// 32-row tiles for the sequence-resident attention kernels (b4r_attn32.hip): v_mfma_f32_32x32x16_bf16 operands as bf16 hi / lo
// "panel" images in LDS, 32 rows x 32 columns x 2 bytes = 2 KB per image, hi image then lo image = one 4 KB tile.
//
// v_mfma_f32_32x32x16_bf16, lane l = 32 h + r:   A[row r][k = 8h + j],  B[k = 8h + j][col r]  (j = 0..7 in the lane's fragment),
//                                                 D[row (t & 3) + 8 (t >> 2) + 4h][col r] in accumulator register t = 0..15.
// An accumulator tile X[rows][cols] is the B operand of a following product that sums over X's rows: registers 8s .. 8s+7 (as
// bf16 hi / lo) are the fragment of k-step s, and hardware k-slot (h, j) then carries X's row 16s + 8 (j >> 2) + 4h + (j & 3).
// The other operand has to present the same row in the same slot:
//   * read by ROWS from a panel image it is chunk 2s + h of row r -- if the image was WRITTEN from an accumulator whose rows are the
//     image's columns (lane (r, h) stores registers 8s .. 8s+7 as chunk 2s + h of row r: acc_to_rows).  Such an image holds matrix
//     column 16s + 8a + 4h' + b at column position 16s + 8h' + 4a + b (bits 2 and 3 swapped): "swapped" column order.
//   * read TRANSPOSED (ds_read_b64_tr_b16) from an image whose rows are the summation index it is two 4-row blocks, rows
//     16s + 4h + 8jj + (0..3): tr_perm.  tr_nat reads rows 16s + 8h + 4jj + (0..3): the natural slot order of two LDS operands.
// A transposed read delivers image COLUMN 16 cb + i (cb = (l >> 4) & 1, i = l & 15) = MFMA row / column r: the output rows of a
// product whose A operand is a transposed read are the image's column positions.  With an image in swapped column order the output
// register t of lane half h is then matrix column 16 (t >> 3) + 8h + (t & 7): eight consecutive columns per k-step -- the natural
// order a following row read or a 32-byte global store wants.
//
// Bank conflicts: 16-byte chunk c of row `row` sits at chunk position c ^ ((row >> 2) & 3).  ds_read_b128 serves lanes
// {0-3, 12-15, 20-27} together: rows congruent mod 4 then differ in (row >> 2) & 3; a transposed read covers four whole rows.
#pragma once
#include "b4r_block_tiles.h"

namespace {

constexpr int P_IMG = 2048;    // one 32 x 32 bf16 image
constexpr int P_TILE = 4096;   // hi image | lo image

__device__ __forceinline__ f32x16 mfma32(const bf16x8 a, const bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32x3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x16 c) {
  c = mfma32(al, bh, c);
  c = mfma32(ah, bl, c);
  c = mfma32(ah, bh, c);
  return c;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int t = 0; t < 16; ++t) z[t] = 0.f;
  return z;
}
__device__ __forceinline__ f32x8 regs8(const f32x16& v, int s) {
  return (f32x8){v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]};
}

// byte offset of chunk c (0..3) of row `row` (0..31) inside one image
__device__ __forceinline__ int p_chunk(int row, int c) { return row * 64 + 16 * (c ^ ((row >> 2) & 3)); }

struct Lane32 {
  int r, h;         // MFMA row / column, k half
  int rowc[2];      // row read: chunk 2s + h of row r
  int trn[2][2];    // transposed read, natural slots:  [s][jj] rows 16s + 8h + 4jj + ..
  int trp[2][2];    // transposed read, accumulator slots: [s][jj] rows 16s + 4h + 8jj + ..
};
__device__ __forceinline__ Lane32 lane32(int lane) {
  Lane32 k;
  k.r = lane & 31; k.h = lane >> 5;
  const int cb = (lane >> 4) & 1, i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    k.rowc[s] = p_chunk(k.r, 2 * s + k.h);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      k.trn[s][jj] = p_chunk(16 * s + 8 * k.h + 4 * jj + qq, 2 * cb + (pp >> 1)) + 8 * (pp & 1);
      k.trp[s][jj] = p_chunk(16 * s + 4 * k.h + 8 * jj + qq, 2 * cb + (pp >> 1)) + 8 * (pp & 1);
    }
  }
  return k;
}

// accumulator (rows = the image's columns, lane = the image's row) -> rows of a tile, swapped column order
__device__ __forceinline__ void acc_to_rows(char* tile, const Lane32& lk, const f32x16& v) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 hi, lo;
    split8(regs8(v, s), hi, lo);
    *reinterpret_cast<bf16x8*>(tile + lk.rowc[s]) = hi;
    *reinterpret_cast<bf16x8*>(tile + P_IMG + lk.rowc[s]) = lo;
  }
}
// accumulator registers 8s .. 8s+7 as the B (or A) fragment of k-step s
__device__ __forceinline__ void acc_frag(const f32x16& v, int s, bf16x8& hi, bf16x8& lo) { split8(regs8(v, s), hi, lo); }

// the four floats src[4 (2 g' + h) .. +3], g' = 0..3, of a [32] array in LDS: the values of rows (t & 3) + 8 (t >> 2) + 4h in register t
__device__ __forceinline__ f32x16 rows_of(const float* src, int h) {
  f32x16 v;
#pragma unroll
  for (int gp = 0; gp < 4; ++gp) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(src + 8 * gp + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * gp + e] = q[e];
  }
  return v;
}

// the partner lane's value (lane ^ 32)
__device__ __forceinline__ float other_half(float v, int h) {
  const auto s = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, h ? s[0] : s[1]);
}
__device__ __forceinline__ unsigned other_half_u(unsigned v, int h) {
  const auto s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return h ? s[0] : s[1];
}

// the value of the lane 16 further (lane ^ 16)
__device__ __forceinline__ float other_row(float v) {
  const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  return __builtin_bit_cast(float, (threadIdx.x & 16) ? s[0] : s[1]);
}
// sum over the 16 lanes of a DPP row, the SAME bits in every lane of the row: four exchange steps with involutions (lane pairs,
// pair pairs, mirrored halves, mirrored row), so both partners of a step add the same two numbers.  (Rotations would give every
// lane its own order of the 16 additions: statistics that differ by an ulp between the lanes of one token.)
__device__ __forceinline__ float row_allsum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm:[1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm:[2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // row_mirror
  return v;
}

// a workgroup barrier that waits for the wave's LDS operations only (hipcc's __syncthreads also waits for every outstanding global
// store: 2-3 us of acknowledgements when a wave has just written its results)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// sum over the 32 lanes of a half (h = 0: lanes 0..31, result in lane 31; h = 1: result in lane 63), DPP only
__device__ __forceinline__ float half_sum31(float v) {
  v = row_sum15(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));   // row_bcast:15 into rows 1, 3
  return v;
}

}  // namespace
