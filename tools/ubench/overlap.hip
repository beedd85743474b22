// Micro-benchmark: do bf16 MFMA and VALU work overlap on a gfx950 SIMD (a) inside one wave, (b) across two waves?
// build: hipcc --offload-arch=gfx950 -O3 -o overlap overlap.hip ; run: ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// inline assembly keeps the loop free of compiler-made register shuffles (the builtin version rotated AGPRs every iteration)
#define MFMA(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y))
#define FMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(a), "v"(b))
#define EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r))
template <int MODE>   // 1 mfma, 2 valu, 3 both in one wave, 4 exp only, 5 mfma + exp in one wave
__device__ __forceinline__ void body(float* out, int iters, float a, float b) {
  f32x4 acc[8];
  float v[24];
  bf16x8 x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(a + i); y[i] = (__bf16)(b - i); }
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){a, b, a, b};
  for (int i = 0; i < 24; ++i) v[i] = a * i + threadIdx.x;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 1 || MODE == 3 || MODE == 5) MFMA(acc[i]);
      if (MODE == 2 || MODE == 3) { FMA(v[3 * i]); FMA(v[3 * i + 1]); FMA(v[3 * i + 2]); }
      if (MODE == 4 || MODE == 5) EXP(v[i]);
    }
  }
  asm volatile("s_nop 15\n s_nop 15");
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 24; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_single(float* out, int iters, float a, float b) { body<MODE>(out, iters, a, b); }

// two waves per SIMD: waves 0-3 run MA, waves 4-7 run MB (wave-uniform branch)
template <int MA, int MB>
__global__ __launch_bounds__(512) void k_pair(float* out, int iters, float a, float b) {
  if ((threadIdx.x >> 8) == 0) body<MA>(out, iters, a, b);
  else body<MB>(out, iters, a, b);
}

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
  const int iters = 20000;
  const float a = 1.0001f, b = 0.5f;
#define ONE(M) printf("single wave/SIMD mode %d: %8.1f us  (%.1f cycles/iter at 2.4 GHz)\n", M, t = timeit([&] { hipLaunchKernelGGL(k_single<M>, dim3(256), dim3(256), 0, 0, out, iters, a, b); }), t * 2400.f / iters)
#define TWO(A, B) printf("two waves/SIMD modes %d+%d: %8.1f us  (%.1f cycles/iter)\n", A, B, t = timeit([&] { hipLaunchKernelGGL((k_pair<A, B>), dim3(256), dim3(512), 0, 0, out, iters, a, b); }), t * 2400.f / iters)
  float t;
  ONE(1); ONE(2); ONE(3); ONE(4); ONE(5);
  TWO(1, 1); TWO(2, 2); TWO(1, 2); TWO(1, 4); TWO(3, 3); TWO(4, 4);
  hipFree(out);
  return 0;
}
