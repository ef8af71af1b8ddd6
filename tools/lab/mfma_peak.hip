// Lab: sustained MFMA rate of the chip (power/clock-limited) — NW waves per block, one block per CU x OCC,
// each wave cycles NACC independent 32x32x16 f16 accumulators; no memory traffic in the loop.
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef NACC
#define NACC 4
#endif
extern "C" __global__ void mfma_loop(float* out, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  half8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(0.5f - e * 0.01f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
extern "C" int mfma_launch(float* out, int blocks, int threads, int iters, void* stream) {
  hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
