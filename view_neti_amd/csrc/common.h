// Common device/host helpers for the vneti HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define VNETI_OK 0
#define VNETI_EARG -1
#define VNETI_EUNSUP -2
#define VNETI_EWS -3
#define VNETI_EHIP -4

// thread-local last-error string (see vneti_last_error in api.hip)
void vneti_set_error(const char* fmt, ...);
int vneti_check_launch(const char* what);

#define VN_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      vneti_set_error(__VA_ARGS__);           \
      return VNETI_EARG;                      \
    }                                         \
  } while (0)

// An offset guaranteed to be out of range for any buffer resource we build
// (all tensors are < 2 GiB): raw buffer loads at this offset return zeros.
#define VN_OOB 0x80000000u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t vn_make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 vn_buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
__device__ __forceinline__ u32x2 vn_buf_load8(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
}
__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ u32x4 as_u32x4(half8 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ half4 as_half4(u32x2 v) { return __builtin_bit_cast(half4, v); }
__device__ __forceinline__ u32x2 as_u32x2(half4 v) { return __builtin_bit_cast(u32x2, v); }

__device__ __forceinline__ float vn_silu(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float vn_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float vn_quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float vn_gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
