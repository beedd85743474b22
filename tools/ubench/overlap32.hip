// Micro-benchmark: how much vector work hides behind v_mfma_f32_32x32x16_bf16 on a gfx950 SIMD, inside one wave and across the two
// waves of a SIMD -- the question behind the step structure of b4r_head32.hip (one step = 24 MFMAs + ~140 vector instructions per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o overlap32 overlap32.hip ; run: ./overlap32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y))
#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b))
#define EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r))
#define CVT(d, r, q) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(r), "v"(q))
// MODE bit 0: MFMAs; NV = plain vector instructions per MFMA gap; NE = v_exp per gap; DEP: one accumulator chain (else 4 rotating)
template <int MODE, int NV, int NE, bool DEP>
__device__ __forceinline__ void body(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  float v[32];
  unsigned d = 0;
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a + i); y[i] = (__bf16)(b - i); }
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 16; ++t) acc[i][t] = a + t;
  for (int i = 0; i < 32; ++i) v[i] = a * i + threadIdx.x;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      if (MODE & 1) MFMA(acc[DEP ? 0 : (i & 3)]);
#pragma unroll
      for (int j = 0; j < NV; ++j) FMA(v[(i * NV + j) & 31]);
#pragma unroll
      for (int j = 0; j < NE; ++j) EXP(v[(i + 7 * j) & 31]);
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int t = 0; t < 16; ++t) s += acc[i][t];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + d;
}
template <int MODE, int NV, int NE, bool DEP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, float a, float b) { body<MODE, NV, NE, DEP>(out, iters, a, b); }

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}
int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 2000;
  const float a = 1.0001f, b = 0.5f;
  float t;
#define RUN(MODE, NV, NE, DEP, W) t = timeit([&] { hipLaunchKernelGGL((k<MODE, NV, NE, DEP, W>), dim3(256), dim3(64 * W), 0, 0, out, iters, a, b); }); \
  printf("%d wave(s)/SIMD  mfma %d  valu/gap %d  exp/gap %d  %s: %8.1f us = %7.1f ns per 24-MFMA step\n", W / 4, MODE & 1, NV, NE, DEP ? "one chain " : "four accs ", t, t * 1e3f / iters)
  RUN(1, 0, 0, true, 4); RUN(1, 0, 0, false, 4); RUN(0, 4, 1, true, 4); RUN(0, 6, 1, true, 4);
  RUN(1, 2, 0, true, 4); RUN(1, 4, 0, true, 4); RUN(1, 4, 1, true, 4); RUN(1, 6, 0, true, 4); RUN(1, 6, 1, true, 4); RUN(1, 8, 1, true, 4);
  RUN(1, 0, 0, true, 8); RUN(0, 4, 1, true, 8); RUN(0, 6, 1, true, 8);
  RUN(1, 4, 1, true, 8); RUN(1, 6, 1, true, 8); RUN(1, 8, 1, true, 8);
  hipFree(out);
  return 0;
}
