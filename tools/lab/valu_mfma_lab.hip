// Lab: do VALU work (softmax: v_fma / v_exp / v_cvt) and MFMAs overlap on one SIMD — inside one wave, and across waves?
// Per iteration and wave (the shape of one 64-key x 32-query step of attn_fwd_kernel<40>): 14 v_mfma_f32_32x32x16_f16
// (448 cycles) and 96 full-rate VALU + 32 v_exp_f32.
//   mode 0 MFMA only            1 VALU only
//   mode 2 MFMA block, then VALU block (every wave the same program)
//   mode 3 fine interleave: 1 MFMA, then 7 VALU + 2-3 exp, repeated
//   mode 4 wave-specialised: half the waves run 2x the MFMA block, the other half 2x the VALU block
//   mode 5 as 2, but the odd wave groups start with the VALU block (phase shifted)
//   mode 6 as 2 with s_setprio 1 around the MFMA block
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, c1, c2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

#define MF_BLOCK()                                                                                   \
  { MF(c0, A, B); MF(c1, A, B); MF(c0, A, B); MF(c1, A, B); MF(c0, A, B); MF(c1, A, B); MF(c2, A, B); \
    MF(c3, A, B); MF(c2, A, B); MF(c3, A, B); MF(c2, A, B); MF(c3, A, B); MF(c2, A, B); MF(c3, A, B); }
#define V8(k) { FMA(x[0], k1, k2); FMA(x[1], k1, k2); FMA(x[2], k1, k2); FMA(x[3], k1, k2); FMA(x[4], k1, k2); \
                FMA(x[5], k1, k2); FMA(x[6], k1, k2); FMA(x[7], k1, k2); }
#define E8() { EXP(y[0]); EXP(y[1]); EXP(y[2]); EXP(y[3]); EXP(y[4]); EXP(y[5]); EXP(y[6]); EXP(y[7]); }
#define V_BLOCK() { V8(0); V8(0); V8(0); V8(0); E8(); E8(); V8(0); V8(0); V8(0); V8(0); E8(); E8(); V8(0); V8(0); V8(0); V8(0); }

extern "C" __global__ __launch_bounds__(1024) void vm_lab(float* out, int iters, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  u32x4 A = u32x4{0x3c003c00u, 0x3c003c00u, 0, 0}, B = u32x4{0x3c003c00u, 0, 0x3c003c00u, 0};
  float x[8], y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = 0.001f * (lane + i); y[i] = -0.01f * (lane + i); }
  float k1 = 0.999f, k2 = 0.0001f;
  const int role = (wave >> 2) & 1;
  if (mode == 0) {
    for (int it = 0; it < iters; ++it) MF_BLOCK();
  } else if (mode == 1) {
    for (int it = 0; it < iters; ++it) V_BLOCK();
  } else if (mode == 2) {
    for (int it = 0; it < iters; ++it) { MF_BLOCK(); V_BLOCK(); }
  } else if (mode == 3) {
    for (int it = 0; it < iters; ++it) {
      MF(c0, A, B); V8(0); EXP(y[0]); EXP(y[1]);
      MF(c1, A, B); V8(0); EXP(y[2]); EXP(y[3]);
      MF(c0, A, B); V8(0); EXP(y[4]); EXP(y[5]);
      MF(c1, A, B); V8(0); EXP(y[6]); EXP(y[7]);
      MF(c0, A, B); V8(0); EXP(y[0]); EXP(y[1]);
      MF(c1, A, B); V8(0); EXP(y[2]); EXP(y[3]);
      MF(c2, A, B); V8(0); EXP(y[4]); EXP(y[5]);
      MF(c3, A, B); V8(0); EXP(y[6]); EXP(y[7]);
      MF(c2, A, B); V8(0); EXP(y[0]); EXP(y[1]);
      MF(c3, A, B); V8(0); EXP(y[2]); EXP(y[3]);
      MF(c2, A, B); V8(0); EXP(y[4]); EXP(y[5]);
      MF(c3, A, B); V8(0); EXP(y[6]); EXP(y[7]);
      MF(c2, A, B); EXP(y[0]); EXP(y[1]); EXP(y[2]); EXP(y[3]);
      MF(c3, A, B); EXP(y[4]); EXP(y[5]); EXP(y[6]); EXP(y[7]);
    }
  } else if (mode == 4) {
    if (role == 0) for (int it = 0; it < iters; ++it) { MF_BLOCK(); MF_BLOCK(); }
    else for (int it = 0; it < iters; ++it) { V_BLOCK(); V_BLOCK(); }
  } else if (mode == 5) {
    if (role == 0) for (int it = 0; it < iters; ++it) { MF_BLOCK(); V_BLOCK(); }
    else for (int it = 0; it < iters; ++it) { V_BLOCK(); MF_BLOCK(); }
  } else if (mode == 6) {
    for (int it = 0; it < iters; ++it) {
      asm volatile("s_setprio 1"); MF_BLOCK(); asm volatile("s_setprio 0"); V_BLOCK();
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i] + c2[i] + c3[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += x[i] + y[i];
  if (acc == 123.456f) out[threadIdx.x] = acc;
}
extern "C" void vm_launch(float* out, int blocks, int threads, int iters, int mode, void* stream) {
  hipLaunchKernelGGL(vm_lab, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, out, iters, mode);
}
