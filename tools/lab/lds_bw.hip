// Lab: LDS fragment-read throughput with the GEMM's access pattern (ds_read_b128, 128-B rows, XOR swizzle).
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
#ifndef MODE
#define MODE 0  // 0: swizzled fragment pattern, 1: lane-linear (fully contiguous 1 KB per wave instr), 2: same address (broadcast)
#endif
extern "C" __global__ void lds_loop(unsigned* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  for (int i = threadIdx.x * 16; i < 65536; i += blockDim.x * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 31, fhalf = lane >> 5;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        int row = ((wave * 64 + f * 32 + frow) + it * 8) & 511;
        int off;
        if (MODE == 0) off = lds_off(row, ks * 2 + fhalf);
        else if (MODE == 1) off = ((((wave * 16 + f * 4 + ks) * 1024) + lane * 16) + it * 64) & 65535 & ~15;
        else off = (wave * 1024 + f * 256 + ks * 64 + it * 16) & 65535 & ~15;
        u32x4 v = *reinterpret_cast<const u32x4*>(smem + off);
        acc ^= v;
      }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = acc[0];
}
extern "C" int lds_launch(unsigned* out, int blocks, int threads, int iters, void* stream) {
  hipLaunchKernelGGL(lds_loop, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
