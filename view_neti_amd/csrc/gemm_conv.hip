// MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T )          ("NT" form, f16 in, f32 accumulate)
//
// A is either a plain row-major matrix or an *implicit* im2col view of an NHWC
// image (3x3 taps, stride 1/2, optional fused nearest-2x upsample, optional
// transposed gather for dgrad).  This one kernel family carries every dense
// contraction of the frozen SD networks on the TI train step:
//   - UNet/VAE ResnetBlock2D conv1/conv2, Downsample2D, Upsample2D  (reference: diffusers
//     modules driven from training/coach.py:165-169,197-198)
//   - every Linear / 1x1 conv of BasicTransformerBlock, Transformer2DModel, CLIPEncoder
//     (the to_q/to_k/to_v/to_out calls of models/xti_attention_processor.py:30-55)
//   - their input-gradient (dgrad) passes: same kernel, pre-transposed weights.
//
// Structure (round-1 version): 256-thread workgroup = 4 waves, each wave owns a
// WM x WN sub-tile built from v_mfma_f32_32x32x16_f16; BK = 64; global -> VGPR
// (buffer_load_dwordx4, OOB => 0 gives conv zero padding for free) -> LDS
// (XOR-swizzled, conflict-free ds_read_b128) double buffered with the next
// tile's global loads in flight under the MFMAs; epilogue staged through LDS so
// HBM stores are full 16-byte rows with bias / time-embedding / residual fused.
#include "common.h"
#include "../../include/vneti.h"

namespace {

struct GemmArgs {
  const half_t* A;
  const half_t* B;
  void* C;
  const float* bias;
  const half_t* rowadd;
  const void* resid;
  long long lda, ldb, ldc, ld_rowadd, ldr;
  long long strideA, strideB, strideC;
  uint32_t a_bytes, b_bytes;
  int M, N, K;
  int rows_per_group;
  float alpha;
  int act;
  // implicit conv
  int conv_mode;  // 0 plain, 1 forward gather, 2 transposed gather (dgrad)
  int Hi, Wi, Ci, Ho, Wo, stride, pad_t, pad_l, ups;
  long long ldx;
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

// LDS tile of R rows x 64 halfs (128 B rows), 16-byte chunks XOR-swizzled so that a
// ds_read_b128 by 16 lanes on consecutive rows touches 16 distinct 16-B slots.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BM, int BN, int WM, int WN, bool F32OUT>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_kernel(GemmArgs g) {
  constexpr int NWM = BM / WM, NWN = BN / WN;
  constexpr int NT = NWM * NWN * 64;
  constexpr int A_IT = BM * 8 / NT, B_IT = BN * 8 / NT;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int CS_LD = F32OUT ? (BN + 4) : (BN + 8);  // elements
  constexpr int CS_BYTES = BM * CS_LD * (F32OUT ? 4 : 2);
  constexpr int LDS_BYTES = (2 * STAGE_BYTES > CS_BYTES) ? 2 * STAGE_BYTES : CS_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / NWN) * WM;
  const int wn0 = (wave % NWN) * WN;

  // XCD-aware tile mapping: consecutive ids on one XCD sweep N for a fixed M panel.
  int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / g.tiles_n;
  const int tile_n = bid - tile_m * g.tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;
  const int bz = blockIdx.y;

  const half_t* Ab = g.A + (long long)bz * g.strideA;
  const half_t* Bb = g.B + (long long)bz * g.strideB;
  __amdgpu_buffer_rsrc_t rsA = vn_make_rsrc(Ab, g.a_bytes);
  __amdgpu_buffer_rsrc_t rsB = vn_make_rsrc(Bb, g.b_bytes);

  const int lrow = tid >> 3;  // 0..NT/8-1
  const int lchk = tid & 7;

  // ---- per-thread A row bookkeeping -------------------------------------------------
  uint32_t a_base[A_IT];  // plain: byte offset of row start (+chunk); conv: unused
  int a_py[A_IT], a_px[A_IT], a_pb[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + lrow + (NT / 8) * i;
    bool ok = m < g.M;
    if (g.conv_mode == 0) {
      a_base[i] = ok ? (uint32_t)((long long)m * g.lda * 2 + lchk * 16) : VN_OOB;
      a_py[i] = a_px[i] = a_pb[i] = 0;
    } else {
      int hw = g.Ho * g.Wo;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / g.Wo;
      int ox = rem - oy * g.Wo;
      a_pb[i] = ok ? b : -1;
      if (g.conv_mode == 1) {
        a_py[i] = oy * g.stride - g.pad_t;
        a_px[i] = ox * g.stride - g.pad_l;
      } else {
        a_py[i] = oy + g.pad_t;
        a_px[i] = ox + g.pad_l;
      }
      a_base[i] = 0;
    }
  }
  uint32_t b_base[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int n = n0 + lrow + (NT / 8) * i;
    b_base[i] = (n < g.N) ? (uint32_t)((long long)n * g.ldb * 2 + lchk * 16) : VN_OOB;
  }

  u32x4 ra[A_IT], rb[B_IT];

  auto issue_loads = [&](int kt) {
    const int k0 = kt * 64;
    if (g.conv_mode == 0) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        uint32_t off = (a_base[i] == VN_OOB) ? VN_OOB : a_base[i] + (uint32_t)k0 * 2;
        ra[i] = vn_buf_load16(rsA, off);
      }
    } else {
      const int tap = k0 / g.Ci;
      const int ci0 = k0 - tap * g.Ci;
      const int dy = tap / 3;
      const int dx = tap - dy * 3;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        int iy, ix;
        bool ok = a_pb[i] >= 0;
        if (g.conv_mode == 1) {
          iy = a_py[i] + dy;
          ix = a_px[i] + dx;
          if (g.ups) {
            ok = ok && (unsigned)iy < (unsigned)(2 * g.Hi) && (unsigned)ix < (unsigned)(2 * g.Wi);
            iy >>= 1;
            ix >>= 1;
          } else {
            ok = ok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
          }
        } else {
          int ty = a_py[i] - dy;
          int tx = a_px[i] - dx;
          ok = ok && ty >= 0 && tx >= 0;
          if (g.stride == 2) {
            ok = ok && ((ty | tx) & 1) == 0;
            ty >>= 1;
            tx >>= 1;
          }
          iy = ty;
          ix = tx;
          ok = ok && iy < g.Hi && ix < g.Wi;
        }
        uint32_t off = ok ? (uint32_t)((((long long)a_pb[i] * g.Hi + iy) * g.Wi + ix) * g.ldx * 2 +
                                       (ci0 + lchk * 8) * 2)
                          : VN_OOB;
        ra[i] = vn_buf_load16(rsA, off);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      uint32_t off = (b_base[i] == VN_OOB) ? VN_OOB : b_base[i] + (uint32_t)k0 * 2;
      rb[i] = vn_buf_load16(rsB, off);
    }
  };

  auto store_lds = [&](int buf) {
    char* As = smem + buf * STAGE_BYTES;
    char* Bs = As + BM * 128;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int r = lrow + (NT / 8) * i;
      *reinterpret_cast<u32x4*>(As + lds_off(r, lchk)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int r = lrow + (NT / 8) * i;
      *reinterpret_cast<u32x4*>(Bs + lds_off(r, lchk)) = rb[i];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = g.K / 64;
  issue_loads(0);
  store_lds(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fhalf = lane >> 5;

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue_loads(kt + 1);
    const char* As = smem + cur * STAGE_BYTES;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int r = wm0 + i * 32 + frow;
        af[i] = as_half8(*reinterpret_cast<const u32x4*>(As + lds_off(r, ks * 2 + fhalf)));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        int r = wn0 + j * 32 + frow;
        bf[j] = as_half8(*reinterpret_cast<const u32x4*>(Bs + lds_off(r, ks * 2 + fhalf)));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          // operands swapped: D[row = n][col = m]  => each lane owns 4 consecutive n of one m
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_lds(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue phase 1: acc -> (alpha, bias, act) -> LDS tile Cs[BM][CS_LD] ---------
  // (the trailing __syncthreads of the K loop guarantees nobody still reads the stages)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int ml = wm0 + i * 32 + frow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = wn0 + j * 32 + 8 * q + 4 * fhalf;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[i][j][4 * q + e] * g.alpha;
          int n = n0 + nl + e;
          if (g.bias != nullptr && n < g.N) x += g.bias[n];
          v[e] = apply_act(x, g.act);
        }
        if constexpr (F32OUT) {
          f32x4 o = {v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(smem + ((size_t)ml * CS_LD + nl) * 4) = o;
        } else {
          half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          *reinterpret_cast<half4*>(smem + ((size_t)ml * CS_LD + nl) * 2) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- epilogue phase 2: coalesced row-major stores with fused row-add / residual ------
  if constexpr (F32OUT) {
    float* Cb = reinterpret_cast<float*>(g.C) + (long long)bz * g.strideC;
    const float* Rb = reinterpret_cast<const float*>(g.resid);
    if (Rb) Rb += (long long)bz * g.strideC;
    constexpr int CPR = BN / 4;
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      int r = idx / CPR, c = (idx - r * CPR) * 4;
      int m = m0 + r, n = n0 + c;
      if (m >= g.M || n >= g.N) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((size_t)r * CS_LD + c) * 4);
      if (n + 4 <= g.N) {
        if (Rb) {
          f32x4 rr = *reinterpret_cast<const f32x4*>(Rb + (long long)m * g.ldr + n);
          v += rr;
        }
        *reinterpret_cast<f32x4*>(Cb + (long long)m * g.ldc + n) = v;
      } else {
        for (int e = 0; e < 4 && n + e < g.N; ++e) {
          float x = v[e];
          if (Rb) x += Rb[(long long)m * g.ldr + n + e];
          Cb[(long long)m * g.ldc + n + e] = x;
        }
      }
    }
  } else {
    half_t* Cb = reinterpret_cast<half_t*>(g.C) + (long long)bz * g.strideC;
    const half_t* Rb = reinterpret_cast<const half_t*>(g.resid);
    if (Rb) Rb += (long long)bz * g.strideC;
    constexpr int CPR = BN / 8;
    for (int idx = tid; idx < BM * CPR; idx += NT) {
      int r = idx / CPR, c = (idx - r * CPR) * 8;
      int m = m0 + r, n = n0 + c;
      if (m >= g.M || n >= g.N) continue;
      half8 v = as_half8(*reinterpret_cast<const u32x4*>(smem + ((size_t)r * CS_LD + c) * 2));
      const half_t* radd = g.rowadd ? g.rowadd + (long long)(m / g.rows_per_group) * g.ld_rowadd + n : nullptr;
      if (n + 8 <= g.N) {
        if (radd) {
          half8 t = *reinterpret_cast<const half8*>(radd);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)t[e]);
        }
        if (Rb) {
          half8 rr = *reinterpret_cast<const half8*>(Rb + (long long)m * g.ldr + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr[e]);
        }
        *reinterpret_cast<half8*>(Cb + (long long)m * g.ldc + n) = v;
      } else {
        for (int e = 0; e < 8 && n + e < g.N; ++e) {
          float x = (float)v[e];
          if (radd) x = (float)(half_t)(x + (float)radd[e]);
          if (Rb) x += (float)Rb[(long long)m * g.ldr + n + e];
          Cb[(long long)m * g.ldc + n + e] = (half_t)x;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(GemmArgs& g, int batch, bool f32out, hipStream_t st) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, BN);
  dim3 grid(g.tiles_m * g.tiles_n, batch, 1);
  dim3 block((BM / WM) * (BN / WN) * 64);
  if (f32out)
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true>), grid, block, 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false>), grid, block, 0, st, g);
  return vneti_check_launch("gemm_kernel");
}

// tile heuristic: fill >= ~1.5 waves of the 256 CUs when possible, prefer the bigger tile
int select_tile(int M, int N, int batch) {
  long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * batch;
  long long t12864 = (long long)cdiv(M, 128) * cdiv(N, 64) * batch;
  if (N <= 64) return (M >= 2048) ? 2 : 3;
  if (t128 >= 384) return 1;
  if (t12864 >= 384) return 2;
  return 3;
}

}  // namespace

extern "C" int vneti_gemm_select_tile(int M, int N, int batch) { return select_tile(M, N, batch > 0 ? batch : 1); }

extern "C" int vneti_gemm_f16(const vneti_gemm_desc* d, void* stream) {
  VN_REQUIRE(d != nullptr, "gemm: null descriptor");
  VN_REQUIRE(d->A && d->B && d->C, "gemm: null operand pointer");
  VN_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  VN_REQUIRE(d->K % 64 == 0, "gemm: K=%d must be a multiple of 64", d->K);
  VN_REQUIRE(d->ldb % 8 == 0, "gemm: ldb=%lld must be a multiple of 8", d->ldb);
  int batch = d->batch > 0 ? d->batch : 1;
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = (const half_t*)d->A;
  g.B = (const half_t*)d->B;
  g.C = d->C;
  g.bias = d->bias;
  g.rowadd = (const half_t*)d->rowadd;
  g.resid = d->resid;
  g.lda = d->lda;
  g.ldb = d->ldb;
  g.ldc = d->ldc;
  g.ld_rowadd = d->ld_rowadd;
  g.ldr = d->ldr;
  g.strideA = d->strideA;
  g.strideB = d->strideB;
  g.strideC = d->strideC;
  g.M = d->M;
  g.N = d->N;
  g.K = d->K;
  g.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1;
  g.alpha = d->alpha;
  g.act = d->act;
  g.conv_mode = d->conv_mode;
  long long a_bytes;
  if (d->conv_mode == 0) {
    VN_REQUIRE(d->lda % 8 == 0, "gemm: lda=%lld must be a multiple of 8", d->lda);
    a_bytes = ((long long)(d->M - 1) * d->lda + d->K) * 2;
  } else {
    VN_REQUIRE(d->conv_mode == 1 || d->conv_mode == 2, "gemm: bad conv_mode %d", d->conv_mode);
    VN_REQUIRE(d->Ci > 0 && d->Ci % 64 == 0, "conv: Ci=%d must be a multiple of 64", d->Ci);
    VN_REQUIRE(d->K == 9 * d->Ci, "conv: K=%d must equal 9*Ci=%d", d->K, 9 * d->Ci);
    VN_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
    VN_REQUIRE(d->Ho > 0 && d->Wo > 0 && d->M % (d->Ho * d->Wo) == 0, "conv: M=%d not a multiple of Ho*Wo", d->M);
    VN_REQUIRE(d->ldx % 8 == 0, "conv: ldx must be a multiple of 8");
    VN_REQUIRE(!(d->ups && d->conv_mode != 1), "conv: fused upsample only in forward gather mode");
    g.Hi = d->Hi;
    g.Wi = d->Wi;
    g.Ci = d->Ci;
    g.Ho = d->Ho;
    g.Wo = d->Wo;
    g.stride = d->stride;
    g.pad_t = d->pad_t;
    g.pad_l = d->pad_l;
    g.ups = d->ups;
    g.ldx = d->ldx;
    long long nb = d->M / (d->Ho * d->Wo);
    a_bytes = nb * d->Hi * d->Wi * d->ldx * 2;
  }
  long long b_bytes = ((long long)(d->N - 1) * d->ldb + d->K) * 2;
  VN_REQUIRE(a_bytes < 0x7fffffffLL && b_bytes < 0x7fffffffLL, "gemm: operand larger than 2 GiB");
  g.a_bytes = (uint32_t)a_bytes;
  g.b_bytes = (uint32_t)b_bytes;
  if (!d->out_f32) {
    VN_REQUIRE(d->ldc % 8 == 0 || d->N < 8, "gemm: ldc=%lld must be a multiple of 8", d->ldc);
  }
  hipStream_t st = (hipStream_t)stream;
  bool f32 = d->out_f32 != 0;
  VN_REQUIRE(!(f32 && d->rowadd), "gemm: rowadd is only supported for f16 output");

  int cfg = d->tile_hint;
  if (cfg == 0) cfg = select_tile(d->M, d->N, batch);
  switch (cfg) {
    case 1: return launch_cfg<128, 128, 64, 64>(g, batch, f32, st);
    case 2: return launch_cfg<128, 64, 64, 32>(g, batch, f32, st);
    case 3: return launch_cfg<64, 64, 32, 32>(g, batch, f32, st);
    case 4: return launch_cfg<256, 128, 128, 64>(g, batch, f32, st);
    default: vneti_set_error("gemm: unknown tile_hint %d", cfg); return VNETI_EARG;
  }
}
