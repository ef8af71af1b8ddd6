// Lab: how fast can a CU pull L2-resident bytes into LDS (or registers)?  MODE 0: LDS-DMA (buffer_load_dwordx4 ... lds),
// 1: buffer_load_dwordx4 -> VGPR -> ds_write_b128, 2: buffer_load_dwordx4 -> VGPR only, 3: LDS-DMA dword (b32) pieces.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef MODE
#define MODE 0
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
extern "C" __global__ __launch_bounds__(1024) void fill(const char* src, unsigned* out, int iters, int span_bytes, int per_block_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (size_t)blockIdx.x * per_block_stride, (uint32_t)span_bytes);
  u32x4 acc = {0, 0, 0, 0};
  uint32_t off = (uint32_t)(wave * 1024 + lane * 16);
  const uint32_t step = (uint32_t)nw * 1024u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      uint32_t o = off + (uint32_t)p * step;
      if (o >= (uint32_t)span_bytes) o -= (uint32_t)span_bytes;
      char* dst = smem + ((wave * 8 + p) & 63) * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, o, 0, 0, 0);
      } else if (MODE == 3) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + q * 256), 4, (o & ~1023u) + q * 256 + lane * 4, 0, 0, 0);
      } else {
        u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0));
        if (MODE == 1) *reinterpret_cast<u32x4*>(dst + lane * 16) = v;
        else acc ^= v;
      }
#endif
    }
    off += 8 * step;
    while (off >= (uint32_t)span_bytes) off -= (uint32_t)span_bytes;
    if (MODE == 0 || MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE != 2) __syncthreads();
  }
  if (MODE != 2) acc = *reinterpret_cast<u32x4*>(smem + tid * 16 % 65536);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[tid] = acc[0];
}
extern "C" int fill_launch(const void* src, unsigned* out, int blocks, int threads, int iters, int span, int stride, void* stream) {
  hipLaunchKernelGGL(fill, dim3(blocks), dim3(threads), 65536, (hipStream_t)stream, (const char*)src, out, iters, span, stride);
  return (int)hipGetLastError();
}
