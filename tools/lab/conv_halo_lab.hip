// Lab: 3x3 stride-1 pad-1 implicit-GEMM conv with an LDS-resident halo patch.
// The generic kernel re-fills the A operand once per tap (9x); here the (TH+2)x(TW+2) input patch of a TH x TW
// pixel tile is DMA'd into LDS once per 64-channel chunk and the 9 taps read their fragments from it.  B (weights)
// tiles stream per (tap, chunk) through a 3-stage ring.  NHWC f16 in/out, weights [N][9*Ci] (k = tap*Ci + c).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef TH
#define TH 16
#endif
#ifndef TW
#define TW 16
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
#ifndef PITCH
#define PITCH (TW + 2)
#endif
#define VN_OOB 0x80000000u
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, uint32_t off) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, off, 0, 0, 0);
#endif
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ swz(row)) << 4); }

constexpr int BM = TH * TW;
constexpr int NWM = BM / WM, NWN = BN / WN, NT = NWM * NWN * 64;
constexpr int MI = WM / 32, NI = WN / 32;
constexpr int PROWS = (TH + 2) * PITCH;               // patch rows (pixels), 128 B each
constexpr int PROWS_PAD = (PROWS + 7) / 8 * 8;
constexpr int PATCH_BYTES = PROWS_PAD * 128;
constexpr int P_IT = (PROWS_PAD * 8 + NT - 1) / NT;   // 16-B slots per thread
constexpr int B_STAGE = BN * 128;
constexpr int B_IT = BN * 8 / NT;
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 3 * B_STAGE;

extern "C" __global__ __launch_bounds__(NT) void halo_kernel(const half_t* X, const half_t* Wt, half_t* Y, int Bn, int H,
                                                           int W, int Ci, int N) {
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave / NWN) * WM, wn0 = (wave % NWN) * WN;
  const int tiles_x = W / TW, tiles_y = H / TH, tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  const int tn = bid % tiles_n;
  bid /= tiles_n;
  const int tx0 = (bid % tiles_x) * TW;
  bid /= tiles_x;
  const int ty0 = (bid % tiles_y) * TH;
  const int b = bid / tiles_y;
  const int n0 = tn * BN;
  const long long K = 9LL * Ci;
  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, (int)((long long)Bn * H * W * Ci * 2), 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)((long long)N * K * 2), 0x00020000);

  // ---- patch loader bookkeeping: slot s = tid + NT*i -> patch row pr = s/8, LDS chunk slot s%8 (lane-linear DMA)
  uint32_t p_off[P_IT];
#pragma unroll
  for (int i = 0; i < P_IT; ++i) {
    int s = tid + NT * i;
    int pr = s >> 3, slot = s & 7;
    int py = pr / PITCH, px = pr - py * PITCH;
    int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
    bool ok = pr < PROWS && px < TW + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    int chunk = slot ^ swz(pr);
    p_off[i] = ok ? (uint32_t)((((long long)(b * H + iy) * W + ix) * Ci) * 2 + chunk * 16) : VN_OOB;
  }
  uint32_t b_off[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int s = tid + NT * i;
    int r = s >> 3, slot = s & 7;
    int n = n0 + r;
    int chunk = slot ^ swz(r);
    b_off[i] = n < N ? (uint32_t)((long long)n * K * 2 + chunk * 16) : VN_OOB;
  }
  auto issue_patch = [&](int c0, int buf) {
    char* P = smem + buf * PATCH_BYTES;
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      int s = tid + NT * i;
      if (s < PROWS_PAD * 8) dma16(rsX, P + (wave * 8 + (NT / 8) * i) * 128, p_off[i] == VN_OOB ? VN_OOB : p_off[i] + c0 * 2);
    }
  };
  auto issue_b = [&](int kt, int stage, int nchunks) {  // kt = chunk*9 + tap
    int chunk = kt / 9, tap = kt - chunk * 9;
    uint32_t koff = (uint32_t)((tap * Ci + chunk * 64) * 2);
    char* Bs = smem + 2 * PATCH_BYTES + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) dma16(rsW, Bs + (wave * 8 + (NT / 8) * i) * 128, b_off[i] == VN_OOB ? VN_OOB : b_off[i] + koff);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  // patch row of this lane's A-fragment row for MFMA tile i (tap (0,0)): tile row r -> (ty, tx)
  int a_pr[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    int r = wm0 + i * 32 + frow;
    a_pr[i] = (r / TW) * PITCH + (r % TW);
  }
  const int nchunks = Ci / 64;
  const int nkt = nchunks * 9;
  constexpr int PER_B = B_IT;

  half8 af[2][MI], bf[2][NI];
  auto load_frags = [&](int buf, int pbuf, int stage, int tapoff, int ks) {
#ifdef ABL_NOLDS
    for (int i = 0; i < MI; ++i) for (int e = 0; e < 8; ++e) af[buf][i][e] = (half_t)(float)(lane + ks + i + tapoff);
    for (int j = 0; j < NI; ++j) for (int e = 0; e < 8; ++e) bf[buf][j][e] = (half_t)(float)(lane - ks + j);
    return;
#endif
    const char* P = smem + pbuf * PATCH_BYTES;
    const char* Bs = smem + 2 * PATCH_BYTES + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < MI; ++i) af[buf][i] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(P + lds_off(a_pr[i] + tapoff, ks * 2 + fhalf)));
#pragma unroll
    for (int j = 0; j < NI; ++j) bf[buf][j] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(Bs + lds_off(wn0 + j * 32 + frow, ks * 2 + fhalf)));
  };
  auto mma = [&](int buf) {
#ifndef ABL_NOMFMA
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[buf][j], af[buf][i], acc[i][j], 0, 0, 0);
#else
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[buf][i]));
#pragma unroll
    for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(bf[buf][j]));
#endif
  };
  half8 af4[4][MI], bf4[4][NI];
  auto load_frags4 = [&](int set, int pbuf, int stage, int tapoff, int ks) {
    const char* P = smem + pbuf * PATCH_BYTES;
    const char* Bs = smem + 2 * PATCH_BYTES + stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < MI; ++i) af4[set][i] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(P + lds_off(a_pr[i] + tapoff, ks * 2 + fhalf)));
#pragma unroll
    for (int j = 0; j < NI; ++j) bf4[set][j] = __builtin_bit_cast(half8, *reinterpret_cast<const u32x4*>(Bs + lds_off(wn0 + j * 32 + frow, ks * 2 + fhalf)));
  };
  auto mma4 = [&](int set) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf4[set][j], af4[set][i], acc[i][j], 0, 0, 0);
  };
  auto tap_off = [&](int kt) { int tap = kt % 9; int dy = tap / 3, dx = tap - dy * 3; return dy * PITCH + dx; };

  // prologue: patch 0, B stages 0 and 1
  issue_patch(0, 0);
  issue_b(0, 0, nchunks);
  if (nkt > 1) issue_b(1, 1, nchunks);
  if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_B) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#ifdef FULLSET
  load_frags4(0, 0, 0, tap_off(0), 0);
#else
  load_frags(0, 0, 0, tap_off(0), 0);
#endif
  int st = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const int chunk = kt / 9, tap = kt - chunk * 9;
    const int pbuf = chunk & 1;
    const int st_next = st == 2 ? 0 : st + 1;
    const int st_new = st_next == 2 ? 0 : st_next + 1;
    const int toff = tap_off(kt);
    // next chunk's patch: issued at tap 0 so that the per-step vmcnt waits flush it within a step or two
#ifndef ABL_NODMA
    if (tap == 0 && chunk + 1 < nchunks) issue_patch((chunk + 1) * 64, pbuf ^ 1);
    if (kt + 2 < nkt) issue_b(kt + 2, st_new, nchunks);
#endif
#ifdef FULLSET
    // all four k16 fragment sets of the step in registers: the LDS latency is paid once per step
    load_frags4(1, pbuf, st, toff, 1);
    load_frags4(2, pbuf, st, toff, 2);
    load_frags4(3, pbuf, st, toff, 3);
    mma4(0);
    mma4(1);
    mma4(2);
#ifndef ABL_NODMA
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_B) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
#ifndef ABL_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
    if (kt + 1 < nkt) {
      const int nchunk = (kt + 1) / 9;
      load_frags4(0, nchunk & 1, st_next, tap_off(kt + 1), 0);
    }
    mma4(3);
#else
    load_frags(1, pbuf, st, toff, 1);
    mma(0);
    load_frags(0, pbuf, st, toff, 2);
    mma(1);
    load_frags(1, pbuf, st, toff, 3);
    mma(0);
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER_B) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef ABL_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
    if (kt + 1 < nkt) {
      const int nchunk = (kt + 1) / 9;
      load_frags(0, nchunk & 1, st_next, tap_off(kt + 1), 0);
    }
    mma(1);
#endif
    st = st_next;
  }
  // epilogue (lab): direct stores; tile row r -> pixel (ty0 + r/TW, tx0 + r%TW)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      int r = wm0 + i * 32 + frow;
      long long m = ((long long)(b * H + ty0 + r / TW) * W + tx0 + r % TW);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int n = n0 + wn0 + j * 32 + 8 * q + 4 * fhalf;
        if (n + 4 <= N) {
          half4 o = {(half_t)acc[i][j][4 * q], (half_t)acc[i][j][4 * q + 1], (half_t)acc[i][j][4 * q + 2], (half_t)acc[i][j][4 * q + 3]};
          *reinterpret_cast<half4*>(Y + m * N + n) = o;
        }
      }
    }
}
extern "C" int halo_launch(const void* X, const void* Wt, void* Y, int Bn, int H, int W, int Ci, int N, void* stream) {
  if (H % TH || W % TW || Ci % 64) return -1;
  int tiles = Bn * (H / TH) * (W / TW) * ((N + BN - 1) / BN);
  hipLaunchKernelGGL(halo_kernel, dim3(tiles), dim3(NT), 0, (hipStream_t)stream, (const half_t*)X, (const half_t*)Wt, (half_t*)Y,
                     Bn, H, W, Ci, N);
  return (int)hipGetLastError();
}
