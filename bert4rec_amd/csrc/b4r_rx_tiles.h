// Shared pieces of the split-precision (bf16x3) kernels that sweep 16-row tiles held in LDS as bf16 hi / lo images
// (b4r_attn_rx.hip: attention; b4r_head_rx.hip: the masked-LM head).  See b4r_attn_rx.hip for the layout notes.
#pragma once
#include "b4r_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {


constexpr int WAVES = 8;              // waves per workgroup
constexpr int ROWS_WG = 16 * WAVES;   // queries (keys) per workgroup
constexpr int IMG_BYTES = 16 * 64;       // 16 rows of one image
constexpr int TILE_BYTES = 4 * IMG_BYTES;  // one 16-row tile of the four interleaved images

__device__ __forceinline__ f32x4 mfma_bf(const bf16x8 a, const bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma3(const bf16x8 ah, const bf16x8 al, const bf16x8 bh, const bf16x8 bl, f32x4 c) {
  c = mfma_bf(al, bh, c);
  c = mfma_bf(ah, bl, c);
  c = mfma_bf(ah, bh, c);
  return c;
}
__device__ __forceinline__ void split8(const f32x8 x, bf16x8& hi, bf16x8& lo) { b4r_split8(x, hi, lo); }
__device__ __forceinline__ f32x8 load8(const float* ptr) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(ptr);
  const f32x4 b = *reinterpret_cast<const f32x4*>(ptr + 4);
  return (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ f32x8 cat(const f32x4 a, const f32x4 b) {
  return (f32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// byte offset of 16-byte chunk `ch` (0..3) of row `row` of image 0; image k of the same tile is IMG_BYTES * k further.
// TB = bytes per 16-row tile (number of interleaved images x IMG_BYTES): 4 images in attention, 2 per 32 columns of the
// hidden size in the masked-LM head.
template <int TB = TILE_BYTES>
__device__ __forceinline__ int img_off(int row, int ch) {
  return (row >> 4) * TB + (row & 15) * 64 + 16 * (ch ^ ((0 - (row >> 2)) & 3));
}

// rows [0,nrows) of two [*,32] fp32 head slices -> their bf16 hi / lo images; rows beyond `valid` are zero.  All loads are
// issued before the first conversion (clamped addresses, no guard: a guarded load costs a branch and a full vmcnt(0) round
// trip per iteration, which made this phase half of the kernel time); nrows <= 256 = 4 pieces per thread and tensor.
// NIT = pieces per thread and tensor (4 covers 256 rows with the 512 threads of a workgroup)
template <int NIT = 4> struct StagedRowsT { f32x4 v0[NIT], v1[NIT]; };
typedef StagedRowsT<4> StagedRows;
template <int NIT = 4>
__device__ __forceinline__ void stage_fetch(StagedRowsT<NIT>& st, const float* src0, int ld0, const float* src1, int ld1,
                                            int64_t row0, int valid) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int f = threadIdx.x + 64 * WAVES * it, r = min(f >> 3, valid - 1), c4 = f & 7;
    st.v0[it] = *reinterpret_cast<const f32x4*>(src0 + (row0 + r) * ld0 + 4 * c4);
    st.v1[it] = *reinterpret_cast<const f32x4*>(src1 + (row0 + r) * ld1 + 4 * c4);
  }
}
// images: tensor 0 -> (0 = hi, 1 = lo), tensor 1 -> (2 = hi, 3 = lo)
template <int NIT = 4, int TB = TILE_BYTES>
__device__ __forceinline__ void stage_write(const StagedRowsT<NIT>& st, char* img, int nrows, int valid) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int f = threadIdx.x + 64 * WAVES * it, r = f >> 3, c4 = f & 7;
    if (f < nrows * 8) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 a = r < valid ? st.v0[it] : z, b = r < valid ? st.v1[it] : z;
      bf16x4 ah, al, bh, bl;
      b4r_split4(a, ah, al);
      b4r_split4(b, bh, bl);
      char* dst = img + img_off<TB>(r, c4 >> 1) + 8 * (c4 & 1);
      *reinterpret_cast<bf16x4*>(dst) = ah;
      *reinterpret_cast<bf16x4*>(dst + IMG_BYTES) = al;
      *reinterpret_cast<bf16x4*>(dst + 2 * IMG_BYTES) = bh;
      *reinterpret_cast<bf16x4*>(dst + 3 * IMG_BYTES) = bl;
    }
  }
}

// lane constants of the two fragment reads (tile t adds TILE_BYTES * t)
struct FragAddr {
  int row;     // row fragment: row 16t + (lane & 15), columns 8g .. 8g+7
  int tr[2];   // transposed fragment of column block db: this lane's address of the 4 x 16 block at rows 16t + 4g ..
};
__device__ __forceinline__ FragAddr frag_addr(int lane) {
  const int i = lane & 15, g = lane >> 4, qq = i >> 2, pp = i & 3;
  FragAddr a;
  a.row = img_off(i, g);
#pragma unroll
  for (int db = 0; db < 2; ++db) a.tr[db] = img_off(4 * g + qq, 2 * db + (pp >> 1)) + 8 * (pp & 1);
  return a;
}
// `tile` = image base + lane constant + TILE_BYTES * t, computed once per tile by the caller; IMG = image index
template <int IMG>
__device__ __forceinline__ bf16x8 row_frag(const char* tile) {
  return *reinterpret_cast<const bf16x8*>(tile + IMG * IMG_BYTES);
}
// element j < 4: image[16*t0 + 4g + j][16*db + (lane&15)], element j >= 4: the same of the next tile (t0 + 1)
template <int IMG, int TB = TILE_BYTES>
__device__ __forceinline__ bf16x8 tr_frag(const char* tile) {
  typedef __attribute__((address_space(3))) s16x4* lds_ptr;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + IMG * IMG_BYTES));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + IMG * IMG_BYTES + TB));
  const s16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, r);
}

}  // namespace
