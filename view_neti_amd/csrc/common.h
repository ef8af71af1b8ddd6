// Common device/host helpers for the vneti HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

// The 16-bit storage / MFMA operand type of the whole library.  Default build: IEEE half (`optim.mixed_precision: fp16`,
// training/coach.py:792-794).  -DVN_BF16 (csrc/build.py builds it as libvneti_hip_bf16.so): bfloat16 — accelerate's
// `mixed_precision: bf16` branch of the reference (training/coach.py:796-802).  Same kernels, same f32 accumulation and
// statistics; only the operand format (and therefore the MFMA opcode, which runs at the same rate) differs.
#ifdef VN_BF16
typedef __bf16 half_t;
#define VN_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define VN_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define VN_FDOT2(a, b, c) __builtin_amdgcn_fdot2_f32_bf16((a), (b), (c), false)  /* c + a.x*b.x + a.y*b.y, f32 */
#define VN_PRECISION 1
#else
typedef _Float16 half_t;
#define VN_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define VN_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define VN_FDOT2(a, b, c) __builtin_amdgcn_fdot2((a), (b), (c), false)  /* c + a.x*b.x + a.y*b.y, f32 */
#define VN_PRECISION 0
#endif
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// a + b rounded to the 16-bit storage format, eight channels at a time (the residual / row-add operands of the GEMM
// epilogues).  fp16: four v_pk_add_f16 instead of 8 x (cvt, cvt, add, cvt) — bit-identical to rounding the f32 sum: the f32
// sum of two halfs is exact up to 12 binades apart and beyond that lies on the same side of every f16 rounding boundary
// (checked on 10^8 random and structured pairs).  bf16 has no packed add on gfx950: through f32 as before.
__device__ __forceinline__ half8 vn_add8(const half8 a, const half8 b) {
#ifdef VN_BF16
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (half_t)((float)a[e] + (float)b[e]);
  return r;
#else
  return a + b;
#endif
}
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
// Write-through (sc1) 16-byte store of a kernel's OUTPUT through a buffer resource (base wave-uniform, byte offset < 2 GiB
// like every tensor here).  A plain store leaves the line dirty in the XCD's L2 until the end-of-kernel release writes it back —
// on the critical path between two dependent launches (boundary cost + dirty bytes / 6 TB/s, MI355X_MICROARCH.md price list).
// With sc1 the line goes to the fabric while the kernel is still running; nothing re-reads an output inside the launch that
// wrote it, and the next launch's acquire invalidates L2 anyway.  Round 6, same box, same picks: 38.5 -> 40.7 steps/s
// (profiles/r06_wt_stores_ab*.txt).  The builtin keeps the store visible to the compiler's hazard and waitcnt passes (an
// inline-asm store needed hand-placed wait states and still broke two epilogues).
constexpr int VN_CPOL_NT = 2;    // aux bit 1: non-temporal (streaming) hint
constexpr int VN_CPOL_SC1 = 16;  // aux / cache-policy bit 4: sc1 on gfx940+
template <typename V>
__device__ __forceinline__ void vn_st16_wt(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, const V v) {
  static_assert(sizeof(V) == 16, "16-byte stores only: narrower sc1 stores are one fabric write each");
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, VN_CPOL_SC1);
}
// Tensors that are written in the forward pass and read ONLY by the backward pass, milliseconds later (the GEGLU
// pre-activation p: 0.6 GB per step; a pre-activation stored beside its activated copy, e.g. CLIP fc1), and their single read
// there, carry the non-temporal hint on top: they would only push soon-needed lines out of the caches.  Round 6, same box, same
// picks: +0.3 ... 0.4 % on the step (profiles/r06_nt_saved_ab1/2.txt).
__device__ __forceinline__ u32x4 vn_buf_load16_once(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, VN_CPOL_NT);
}
template <typename V>
__device__ __forceinline__ void vn_st16_wt_saved(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, const V v) {
  static_assert(sizeof(V) == 16, "16-byte stores only");
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, VN_CPOL_SC1 | VN_CPOL_NT);
}
__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }
__device__ __forceinline__ u32x4 as_u32x4(half8 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ half4 as_half4(u32x2 v) { return __builtin_bit_cast(half4, v); }
__device__ __forceinline__ u32x2 as_u32x2(half4 v) { return __builtin_bit_cast(u32x2, v); }

// sigmoid family on the raw transcendental unit: exp(-x) = v_exp_f32(-x * log2 e) and ONE v_rcp_f32 (1 ulp) instead of an
// IEEE division (v_div_scale x2, v_rcp, four fma, v_div_fmas, v_div_fixup: ~10 VALU issues per element in the
// GroupNorm+SiLU passes and the GEMM epilogues); results are rounded to f16 right after, far above the 1-ulp f32 difference.
__device__ __forceinline__ float vn_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float vn_silu(float x) { return x * vn_sigmoid(x); }
__device__ __forceinline__ float vn_quick_gelu(float x) { return x * vn_sigmoid(1.702f * x); }
// Exact (erf) GELU without libm's erff (~40 instructions): Phi(x) = 0.5 * erfc(-x / sqrt 2) with erfc from Abramowitz &
// Stegun 7.1.26 (|error| < 1.5e-7, far below the f16 the result is rounded to), evaluated on the erfc side so the
// negative tail has no 1 - erf cancellation.  exp(-x^2 / 2) is shared with the derivative: gelu'(x) = Phi(x) + x phi(x).
// Raw v_exp_f32 / v_rcp_f32: ~14 VALU operations, two of them transcendental.
__device__ __forceinline__ void vn_gelu_parts(float x, float& cdf, float& xpdf) {
  const float az = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float E = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);  // exp(-x^2 / 2)
  const float erfc_az = poly * t * E;
  cdf = 0.5f * (x >= 0.f ? 2.f - erfc_az : erfc_az);
  xpdf = x * 0.3989422804014327f * E;
}
__device__ __forceinline__ float vn_gelu_erf(float x) {
  float cdf, xpdf;
  vn_gelu_parts(x, cdf, xpdf);
  return x * cdf;
}
__device__ __forceinline__ float vn_gelu_erf_grad(float x) {
  float cdf, xpdf;
  vn_gelu_parts(x, cdf, xpdf);
  return cdf + xpdf;
}

// ---- order-independent accumulation of float partial sums (the GroupNorm statistics) -------------------------------
// Float atomics make a sum depend on arrival order; integers do not.  A partial sum v is added as two 64-bit integers:
//   hi = rint(v * 2^4)                      (coarse, never overflows: |v| < 2^58)
//   lo = rint(v * 2^40)  modulo 2^64        (fine; wraps for |total| >= 2^23, which the decoder undoes from hi)
// so every run — any block order, any atomic order — produces bit-identical totals with 2^-40 absolute resolution over the
// whole float range.  One accumulated quantity = 2 words; a GroupNorm slot entry = [S1.hi, S1.lo, S2.hi, S2.lo].
typedef unsigned long long vn_u64;
// Non-finite input (an f16 overflow upstream): float -> integer conversion of inf / NaN / > 2^63 is undefined, so v is
// clamped to +-2^56 first (NaN -> -2^56): the statistics are then defined garbage, and the overflow is still detected the
// way GradScaler detects it — the inf itself survives ELEMENT-WISE through the normalisation ((inf - mean) * rstd) into
// the loss / the gradient bucket, where grads_check_finite sets found_inf (tests/test_step_gpu.py::test_overflow_skips_step).
// An additive sentinel cannot do better: N sentinels wrap modulo 2^64 for some N <= 64, and "every tile overflowed" is the
// likely case.
__device__ __forceinline__ void vn_fx_encode(float v, vn_u64& hi, vn_u64& lo) {
  v = fminf(fmaxf(v, -7.2057594e16f), 7.2057594e16f);
  const float vh = rintf(v * 16.f);
  const long long h = (long long)vh;
  const float rem = v - vh * 0.0625f;  // exact: |rem| <= 2^-5, and 0 once |v| >= 2^19
  hi = (vn_u64)h;
  lo = ((vn_u64)h << 36) + (vn_u64)(long long)rintf(rem * 1099511627776.f);
}
__device__ __forceinline__ double vn_fx_decode(vn_u64 hi_u, vn_u64 lo_u) {
  const long long hi = (long long)hi_u, lo = (long long)lo_u;
  // total * 2^40 = lo + k * 2^64, k = the integer that brings it next to hi * 2^36
  const double k = rint(((double)hi * 68719476736.0 - (double)lo) * 5.421010862427522e-20);
  return (double)lo * 9.094947017729282e-13 + k * 16777216.0;
}
// add (s, q) to a 4-word entry with integer atomics (LDS or global)
__device__ __forceinline__ void vn_fx_add2(vn_u64* e, float s, float q) {
  vn_u64 h, l;
  vn_fx_encode(s, h, l);
  atomicAdd(e, h);
  atomicAdd(e + 1, l);
  vn_fx_encode(q, h, l);
  atomicAdd(e + 2, h);
  atomicAdd(e + 3, l);
}

// Wave-wide all-reduce without the LDS crossbar: rotations inside the four 16-lane DPP rows (row_ror:8/4/2/1, one VALU
// instruction each), then the four row results through v_readlane.  __shfl_xor is a ds_bpermute per step — six dependent
// LDS round trips — and these reductions sit on the critical path of the one-row-per-wave kernels (LayerNorm, the text
// path).  All 64 lanes must be active (every caller reduces with full waves).
template <int N>
__device__ __forceinline__ float vn_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float vn_readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += vn_row_ror<8>(v);
  v += vn_row_ror<4>(v);
  v += vn_row_ror<2>(v);
  v += vn_row_ror<1>(v);
  return (vn_readlane(v, 0) + vn_readlane(v, 16)) + (vn_readlane(v, 32) + vn_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, vn_row_ror<8>(v));
  v = fmaxf(v, vn_row_ror<4>(v));
  v = fmaxf(v, vn_row_ror<2>(v));
  v = fmaxf(v, vn_row_ror<1>(v));
  return fmaxf(fmaxf(vn_readlane(v, 0), vn_readlane(v, 16)), fmaxf(vn_readlane(v, 32), vn_readlane(v, 48)));
}

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
// n / d for 0 <= n, 0 < d through a float reciprocal (rcp_d = 1 / d, computed once): the float quotient is within one of the
// true one for n < 2^24 and within a few beyond, and the remainder test makes it exact — a 32-bit integer division is
// ~40 dependent instructions, this is 6.  Returns the quotient, leaves the remainder in `rem`.
__device__ __forceinline__ int vn_divmod(int n, int d, float rcp_d, int& rem) {
  int q = (int)((float)n * rcp_d);
  int r = n - q * d;
  while (r < 0) {
    q -= 1;
    r += d;
  }
  while (r >= d) {
    q += 1;
    r -= d;
  }
  rem = r;
  return q;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
