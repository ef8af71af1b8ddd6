// Lab: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = element index; every lane passes an address.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
extern "C" __global__ void tr_probe(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = 0;                                   // uniform address
  else if (mode == 1) addr = (uint32_t)(l * 8);               // lane-linear 8-byte slots
  else if (mode == 2) addr = (uint32_t)(((l & 15) * 64 + (l >> 4) * 8));   // 16 rows of 64 B, group picks 8-B column block
  else addr = (uint32_t)(((l & 3) * 8 + ((l >> 2) & 3) * 128 + (l >> 4) * 512));
  uint32_t base = (uint32_t)(uintptr_t)lds;  // LDS aperture offset is the low bits of the pointer
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + (base & 0xffff)));
  out[l * 4 + 0] = (unsigned short)(v[0] & 0xffff);
  out[l * 4 + 1] = (unsigned short)(v[0] >> 16);
  out[l * 4 + 2] = (unsigned short)(v[1] & 0xffff);
  out[l * 4 + 3] = (unsigned short)(v[1] >> 16);
}
extern "C" int tr_launch(unsigned short* out, int mode, void* stream) {
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, out, mode);
  return (int)hipGetLastError();
}
