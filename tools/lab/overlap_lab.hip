// Lab: do ds_read_b128 fragment reads and MFMAs of ONE wave overlap, and at what waves/SIMD?
// MODE 0 reads only, 1 MFMA only, 2 "read all, wait, multiply all" (what the compiler emits for the GEMM loop),
// 3 software pipeline (reads of sub-step s+1 issued before the MFMAs of sub-step s, lgkmcnt(4) waits),
// 4 independent interleave (MFMAs never wait for the reads; upper bound of the overlap).
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef MODE
#define MODE 0
#endif
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=&v"(dst) : "v"(addr))
#define MF(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

extern "C" __global__ __launch_bounds__(1024) void ovl(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
#ifdef RANDOM_DATA
  // pseudo-random f16 values in about [-2, 2): the multipliers toggle like they do on real activations
  auto rnd = [](unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; x *= 2654435761u; x ^= x >> 15;
                              return (x & 0x83ff83ffu) | 0x3c003c00u ^ ((x >> 3) & 0x04000400u); };
  for (int i = threadIdx.x * 16; i < 65536; i += blockDim.x * 16)
    *reinterpret_cast<u32x4*>(smem + i) = u32x4{rnd(i + 1 + blockIdx.x * 77), rnd(i + 2), rnd(i + 3 + blockIdx.x), rnd(i + 4)};
#else
  for (int i = threadIdx.x * 16; i < 65536; i += blockDim.x * 16) *reinterpret_cast<u32x4*>(smem + i) = u32x4{0x3c003c00u, 0x3c003c00u, 0, 0};
#endif
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 31, fhalf = lane >> 5;
  // per-wave bases: A rows wave*16.., B rows 256+...; sub-step chunk = 2*s + fhalf handled through 4 address registers
  int a0[4], b0[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a0[s] = lds_off((wave & 3) * 64 + frow, 2 * s + fhalf);
    b0[s] = lds_off(256 + (wave >> 2) * 32 + frow, 2 * s + fhalf) & 65535;
  }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  u32x4 A0[2], A1[2], B0[2], B1[2];
#ifdef RANDOM_DATA
  A0[0] = *reinterpret_cast<u32x4*>(smem + lane * 16); A1[0] = *reinterpret_cast<u32x4*>(smem + 1024 + lane * 16);
  B0[0] = *reinterpret_cast<u32x4*>(smem + 2048 + lane * 16); B1[0] = *reinterpret_cast<u32x4*>(smem + 3072 + lane * 16);
  A0[1] = *reinterpret_cast<u32x4*>(smem + 4096 + lane * 16); A1[1] = *reinterpret_cast<u32x4*>(smem + 5120 + lane * 16);
  B0[1] = *reinterpret_cast<u32x4*>(smem + 6144 + lane * 16); B1[1] = *reinterpret_cast<u32x4*>(smem + 7168 + lane * 16);
#else
  A0[0] = A1[0] = B0[0] = B1[0] = A0[1] = A1[1] = B0[1] = B1[1] = u32x4{0x3c003c00u, 0, 0, 0};
#endif
  u32x4 D0, D1, D2, D3;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 d[8] = {};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { RD(A0[0], a0[s], 0); RD(A1[0], a0[s], 4096); RD(B0[0], b0[s], 0); RD(B1[0], b0[s], 4096); }
      WAIT(0);
    } else if (MODE == 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) { MF(c0, A0[0], B0[0]); MF(c1, A0[0], B1[0]); MF(c2, A1[0], B0[0]); MF(c3, A1[0], B1[0]); }
    } else if (MODE == 5) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        MF16(d[0], A0[0], B0[0]); MF16(d[1], A0[0], B1[0]); MF16(d[2], A1[0], B0[0]); MF16(d[3], A1[0], B1[0]);
        MF16(d[4], A0[1], B0[1]); MF16(d[5], A0[1], B1[1]); MF16(d[6], A1[1], B0[1]); MF16(d[7], A1[1], B1[1]);
      }
    } else if (MODE == 2) {
      u32x4 fa0[4], fa1[4], fb0[4], fb1[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { RD(fa0[s], a0[s], 0); RD(fa1[s], a0[s], 4096); RD(fb0[s], b0[s], 0); RD(fb1[s], b0[s], 4096); }
      WAIT(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) { MF(c0, fa0[s], fb0[s]); MF(c1, fa0[s], fb1[s]); MF(c2, fa1[s], fb0[s]); MF(c3, fa1[s], fb1[s]); }
    } else if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1, sn = (s + 1) & 3;
        RD(A0[nxt], a0[sn], 0); RD(A1[nxt], a0[sn], 4096); RD(B0[nxt], b0[sn], 0); RD(B1[nxt], b0[sn], 4096);
        WAIT(4);
        MF(c0, A0[cur], B0[cur]); MF(c1, A0[cur], B1[cur]); MF(c2, A1[cur], B0[cur]); MF(c3, A1[cur], B1[cur]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        RD(D0, a0[s], 0); MF(c0, A0[0], B0[0]); RD(D1, a0[s], 4096); MF(c1, A0[0], B1[0]);
        RD(D2, b0[s], 0); MF(c2, A1[0], B0[0]); RD(D3, b0[s], 4096); MF(c3, A1[0], B1[0]);
      }
      WAIT(0);
    }
  }
  WAIT(0);
  float r = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 8; ++i) r += d[i][i & 3];
  if (MODE == 0 || MODE == 4) r += __uint_as_float(A0[0][0] ^ A1[0][1] ^ B0[0][2] ^ B1[0][3]);
  if (MODE == 4) r += __uint_as_float(D0[0] ^ D1[1] ^ D2[2] ^ D3[3]);
  if (r == 123.456f) out[threadIdx.x] = r;
}
extern "C" int ovl_launch(float* out, int blocks, int threads, int iters, void* stream) {
  hipLaunchKernelGGL(ovl, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
