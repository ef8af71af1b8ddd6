// GEMM main-loop lab: same structure as view_neti_amd/csrc/gemm_conv.hip (plain NT GEMM only),
// with ablation macros.  Build: hipcc -DVARIANT_... ; driven by tools/lab/run_lab.py
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef BM
#define BM 128
#endif
#ifndef BN
#define BN 128
#endif
#ifndef WM
#define WM 64
#endif
#ifndef WN
#define WN 64
#endif
#ifndef STAGES
#define STAGES 2
#endif
#define VN_OOB 0x80000000u
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, off, 0, 0, 0);
#endif
}
extern "C" __global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void lab_kernel(const half_t* A, const half_t* B,
                                                                                  half_t* C, int M, int N, int K) {
  constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64, RSTEP = NT / 8;
  constexpr int A_IT = BM / RSTEP, B_IT = BN / RSTEP, MI = WM / 32, NI = WN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
#ifdef XPF
#ifndef XBK
#define XBK 32
#endif
#ifndef XST
#define XST 3
#endif
  __shared__ __attribute__((aligned(16))) char smem[XST * (BM + BN) * XBK * 2];
#else
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES];
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int tiles_n = (N + BN - 1) / BN, tiles_m = (M + BM - 1) / BM;
  int nblk = tiles_m * tiles_n, bid = blockIdx.x;
  { int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3; bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx; }
  const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((long long)M * K * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((long long)N * K * 2), 0x00020000);
  const int lrow = tid >> 3;
  const int gchunk = (tid & 7) ^ ((lrow >> 1) & 7);
  int a_off[A_IT], b_off[B_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) { int m = m0 + lrow + RSTEP * i; a_off[i] = m < M ? m * K * 2 + gchunk * 16 : -1; }
#pragma unroll
  for (int i = 0; i < B_IT; ++i) { int n = n0 + lrow + RSTEP * i; b_off[i] = n < N ? n * K * 2 + gchunk * 16 : -1; }
  auto issue = [&](int kt, int stage) {
    char* As = smem + stage * STAGE_BYTES; char* Bs = As + BM * 128;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) dma16(rsA, As + (wave * 8 + RSTEP * i) * 128, a_off[i] < 0 ? VN_OOB : (uint32_t)(a_off[i] + kt * 128));
#pragma unroll
    for (int i = 0; i < B_IT; ++i) dma16(rsB, Bs + (wave * 8 + RSTEP * i) * 128, b_off[i] < 0 ? VN_OOB : (uint32_t)(b_off[i] + kt * 128));
  };
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = K / 64;
  const int frow = lane & 31, fhalf = lane >> 5;
  (void)nk;
#ifdef XPF
  // ---- cross-barrier fragment prefetch: XST stages of BK=XBK, the fragments of step s+1 are read from LDS
  // (stage made visible by the previous barrier) under the MFMAs of step s --------------------------------
  {
    constexpr int KSUB = XBK / 16;               // k16 sub-steps per stage
    constexpr int ROWB = XBK * 2;                // bytes per LDS row
    constexpr int CPR = ROWB / 16;               // 16-B chunks per row
    constexpr int XSTAGE = (BM + BN) * ROWB;
    constexpr int RPP = NT / CPR;                // rows per DMA pass of the whole block
    constexpr int XA_IT = BM / RPP, XB_IT = BN / RPP;
    constexpr int RPW = 64 / CPR;                // rows per wave-instruction
    const int xrow = tid / CPR, xslot = tid % CPR;
    auto swz = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
    const int xchunk = xslot ^ swz(xrow);
    int xa[XA_IT], xb[XB_IT];
#pragma unroll
    for (int i = 0; i < XA_IT; ++i) { int m = m0 + xrow + RPP * i; xa[i] = m < M ? m * K * 2 + xchunk * 16 : -1; }
#pragma unroll
    for (int i = 0; i < XB_IT; ++i) { int n = n0 + xrow + RPP * i; xb[i] = n < N ? n * K * 2 + xchunk * 16 : -1; }
    auto xissue = [&](int kt, int stage) {
      char* As = smem + stage * XSTAGE; char* Bs = As + BM * ROWB;
#pragma unroll
      for (int i = 0; i < XA_IT; ++i) dma16(rsA, As + (wave * RPW + RPP * i) * ROWB, xa[i] < 0 ? VN_OOB : (uint32_t)(xa[i] + kt * ROWB));
#pragma unroll
      for (int i = 0; i < XB_IT; ++i) dma16(rsB, Bs + (wave * RPW + RPP * i) * ROWB, xb[i] < 0 ? VN_OOB : (uint32_t)(xb[i] + kt * ROWB));
    };
    auto xoff = [&](int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); };
    constexpr int PER = XA_IT + XB_IT;
    const int nks = K / XBK;
    half8 af[2][MI], bf[2][NI];
    auto xload = [&](int buf, int stage, int ks) {
      const char* As = smem + stage * XSTAGE; const char* Bs = As + BM * ROWB;
#pragma unroll
      for (int i = 0; i < MI; ++i) af[buf][i] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(As + xoff(wm0 + i * 32 + frow, ks * 2 + fhalf)));
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[buf][j] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(Bs + xoff(wn0 + j * 32 + frow, ks * 2 + fhalf)));
    };
    auto xmma = [&](int buf) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[buf][j], af[buf][i], acc[i][j], 0, 0, 0);
    };
    // prologue: XST-1 stages in flight, stage 0 landed + visible, its first fragments in buffer 0
    for (int t = 0; t < XST - 1 && t < nks; ++t) xissue(t, t);
    if (nks > XST - 2 && XST > 2) { if (XST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    xload(0, 0, 0);
    int st = 0;  // LDS slot of the current step
    for (int s = 0; s < nks; ++s) {
      int st_new = st + XST - 1; if (st_new >= XST) st_new -= XST;
      int st_next = st + 1; if (st_next >= XST) st_next -= XST;
      if (s + XST - 1 < nks) xissue(s + XST - 1, st_new);
#pragma unroll
      for (int ks = 0; ks < KSUB; ++ks) {
        if (ks + 1 < KSUB) {
          xload((ks + 1) & 1, st, ks + 1);
          xmma(ks & 1);
        } else {
          // the barrier that publishes stage s+1 sits before the LAST MFMA group, so the first fragments of
          // the next step are fetched under it and the step boundary has no LDS-latency bubble
          const int younger = min(XST - 2, nks - 2 - s);
          if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER) : "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (s + 1 < nks) xload(0, st_next, 0);
          xmma(ks & 1);
        }
      }
      st = st_next;
    }
  }
#elif STAGES == 2
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
#ifndef NOLOAD
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
#endif
    const char* As = smem + cur * STAGE_BYTES; const char* Bs = As + BM * 128;
#ifdef FRAGDB
    half8 afq[2][MI], bfq[2][NI];
#define LOADFR(buf, ks)                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) afq[buf][i] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(As + lds_off(wm0 + i * 32 + frow, (ks) * 2 + fhalf))); \
    _Pragma("unroll") for (int j = 0; j < NI; ++j) bfq[buf][j] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(Bs + lds_off(wn0 + j * 32 + frow, (ks) * 2 + fhalf)));
#define MMA(buf)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)   \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfq[buf][j], afq[buf][i], acc[i][j], 0, 0, 0);
#ifdef SETPRIO
#define PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PRIO(x)
#endif
    LOADFR(0, 0)
    LOADFR(1, 1)
    PRIO(1); MMA(0) PRIO(0);
    LOADFR(0, 2)
    PRIO(1); MMA(1) PRIO(0);
    LOADFR(1, 3)
    PRIO(1); MMA(0) PRIO(0);
    PRIO(1); MMA(1) PRIO(0);
#else
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 af[MI], bf[NI];
#ifndef NOLDS
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(As + lds_off(wm0 + i * 32 + frow, ks * 2 + fhalf)));
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(Bs + lds_off(wn0 + j * 32 + frow, ks * 2 + fhalf)));
#else
#pragma unroll
      for (int i = 0; i < MI; ++i) { for (int e = 0; e < 8; ++e) af[i][e] = (half_t)(float)(lane + ks + i); }
#pragma unroll
      for (int j = 0; j < NI; ++j) { for (int e = 0; e < 8; ++e) bf[j][e] = (half_t)(float)(lane - ks + j); }
#endif
#ifndef NOMFMA
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
#else
#pragma unroll
      for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
      for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(bf[j]));
#endif
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#else  // N-stage ring: loads run STAGES-1 tiles ahead; counted vmcnt, one raw barrier per tile
  constexpr int PER = A_IT + B_IT;
  constexpr int D = STAGES - 1;
  for (int t = 0; t < D && t < nk; ++t) issue(t, t % STAGES);
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(D - 1, nk - 1 - kt);
    switch (ahead) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    if (kt + D < nk) issue(kt + D, (kt + D) % STAGES);
    const int cur = kt % STAGES;
    const char* As = smem + cur * STAGE_BYTES; const char* Bs = As + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(As + lds_off(wm0 + i * 32 + frow, ks * 2 + fhalf)));
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(Bs + lds_off(wn0 + j * 32 + frow, ks * 2 + fhalf)));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
#endif
  // minimal epilogue: direct stores (lab only)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int m = m0 + wm0 + i * 32 + frow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int n = n0 + wn0 + j * 32 + 8 * q + 4 * fhalf;
        if (m < M && n + 4 <= N) {
          half4 o = {(half_t)acc[i][j][4 * q], (half_t)acc[i][j][4 * q + 1], (half_t)acc[i][j][4 * q + 2], (half_t)acc[i][j][4 * q + 3]};
          *reinterpret_cast<half4*>(C + (long long)m * N + n) = o;
        }
      }
    }
}
extern "C" int lab_launch(const void* A, const void* B, void* C, int M, int N, int K, void* stream) {
  int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipLaunchKernelGGL(lab_kernel, dim3(tiles), dim3((BM / WM) * (BN / WN) * 64), 0, (hipStream_t)stream, (const half_t*)A,
                     (const half_t*)B, (half_t*)C, M, N, K);
  return (int)hipGetLastError();
}
