// Argument block and small device helpers shared by the GEMM / implicit-conv kernels (gemm_conv.hip: the generic tile
// family; gemm8.hip: the 256x256 8-phase ping-pong tile).  Everything here is internal to the library; the C ABI is
// vneti_gemm_f16 (include/vneti.h).
#pragma once
#include "common.h"

namespace {

struct GemmArgs {
  const half_t* A;
  const half_t* B;
  void* C;
  const float* bias;
  const half_t* rowadd;
  const void* resid;
  float* ws;
  const half_t* gate_src;
  half_t* C2;
  long long ld_gate, ldc2;
  int gate_act, act2;
  float* gn_sums;  // [image][slot][G][4] 64-bit fixed-point (sum, sum of squares) of the f16 output (common.h vn_fx_*)
  int gn_hw, gn_cpg, gn_G, gn_slots;
  int geglu;  // 1: C2 = h * gelu(g) of the interleaved tile; 2: C[M][2N] = GEGLU backward against gate_src (see vneti.h)
  long long lda, ldb, ldc, ld_rowadd, ldr;
  long long strideA, strideB, strideC;
  uint32_t a_bytes, b_bytes;
  int M, N, K;
  int rows_per_group;
  float alpha;
  int act;
  int out_f32;
  int batch, ksplit, kt_per_split;
  // implicit conv
  int conv_mode;  // 0 plain, 1 forward gather, 2 transposed gather (dgrad)
  int Hi, Wi, Ci, Ho, Wo, stride, pad_t, pad_l, ups;
  int ldx2;  // pixel stride in bytes
  int korder;  // 0: K = (tap, ci); 1: K = (ci / 64, tap, ci % 64)
  int tiles_m, tiles_n;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case 1: return vn_silu(v);
    case 2: return vn_quick_gelu(v);
    case 3: return vn_gelu_erf(v);
    default: return v;
  }
}
// d act(x) / dx
__device__ __forceinline__ float act_grad(float f, int act) {
  if (act == 2) {
    float s = vn_sigmoid(1.702f * f);
    return s * (1.f + 1.702f * f * (1.f - s));
  } else if (act == 3) {
    return vn_gelu_erf_grad(f);
  }
  float s = vn_sigmoid(f);
  return s * (1.f + f * (1.f - s));
}

// LDS tile of R rows x 64 halfs (128 B rows), 16-byte chunks XOR-swizzled so that a
// ds_read_b128 by 16 lanes on consecutive rows touches 16 distinct 16-B slots.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// one LDS-DMA instruction: 64 lanes x 16 B land at lds_dst + lane*16 (lds_dst wave-uniform).
// The builtin only exists in the device compilation pass.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, off, 0, 0, 0);
#else
  (void)rs; (void)lds_dst; (void)off;
#endif
}

}  // namespace

// the 8-phase ping-pong tiles (gemm8.hip): 256 x bn, bn = 256 or 128 (halo: the 16 x 16-pixel halo-patch convolution form
// of the 256 x 128 tile); g.ksplit / g.kt_per_split already set, f16 output only
int vneti_launch_gemm8(void* gemm_args, int bn, int halo, hipStream_t st);
