// Lab (measured rather than sized): a 256x256 tile as FOUR waves — one per SIMD — with 128x128 wave tiles.
// The accumulators (256 f32 per lane) live in the accumulator half of the 512-register file; a k32 sub-step reads 16 fragments
// for 64 MFMAs (256 LDS bytes per MFMA against 384 for the 8-phase tile's 128x64 wave tiles).  A single wave per SIMD cannot
// afford LDS-DMA (60..185 blocking cycles per 1 KiB piece, 16 pieces per wave and K-tile), so staging goes through registers:
// 16 x buffer_load_dwordx4 per thread issued at the top of a K-tile, written to the other LDS stage after its MFMAs.
// Plain NT GEMM, f16 in / f16 out, M, N multiples of 256, K of 64; no epilogue features.   tools/lab/run_gemm4w.py
//
// RESULT (round 5, one MI355X, random operands, interleaved with the product tiles; profiles/r05_gemm4w_lab.txt): bit-equal to
// the 8-phase tile; compiler-scheduled 789-861 TF/s at 4096^3, with the pinned software pipeline below 873-972, against
// 1255-1353 for the 8-phase 256x256 tile (tile_hint 16) and 1265-1435 for hipBLASLt — 0.63x / 0.72x.  What the single
// wave cannot hide: the 16 ds_write_b128 of a K-tile's staging (64 KiB at ~79 B/clk) and the barrier, ~900 cycles against
// 2 048 of MFMA issue; and it has no free issue slots to put them in (an MFMA gap of 16 cycles holds one or two 4-cycle
// instructions, not a 13-cycle store).  The structure needs the hand-placed asm stream the guide describes; plain HIP
// cannot express it, and the prize over the 8-phase tile was sized at a few per cent (profiles/LAB_NOTES.md, round 5).
#include "../../view_neti_amd/csrc/common.h"
#include "../../view_neti_amd/csrc/gemm_args.h"

namespace {
constexpr int BM = 256, BN = 256, TILE_BYTES = 256 * 128;  // one operand tile of a stage: 256 rows x 64 halfs
constexpr int STAGE = 2 * TILE_BYTES;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w_kernel(
    const half_t* __restrict__ A, const half_t* __restrict__ B, half_t* __restrict__ C, int M, int N, int K, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  int nblk = gridDim.x, bid = blockIdx.x;
  {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const __amdgpu_buffer_rsrc_t rsA = vn_make_rsrc(A, (uint32_t)((long long)M * K * 2));
  const __amdgpu_buffer_rsrc_t rsB = vn_make_rsrc(B, (uint32_t)((long long)N * K * 2));

  // staging: thread t moves chunk (t & 7) of rows (t >> 3) + 32 i, i < 8, of both operand tiles
  const int srow = tid >> 3, schunk = tid & 7;
  uint32_t aoff[8], boff[8];
  int soff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = srow + 32 * i;
    aoff[i] = (uint32_t)((long long)(m0 + r) * K * 2) + schunk * 16;
    boff[i] = (uint32_t)((long long)(n0 + r) * K * 2) + schunk * 16;
    soff[i] = lds_off(r, schunk);
  }
  u32x4 ra[8], rb[8];
  auto issue = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[i], kt * 128, 0);
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, boff[i], kt * 128, 0);
    }
  };
  auto commit = [&](int stage) {
    char* s = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<u32x4*>(s + soff[i]) = ra[i];
      *reinterpret_cast<u32x4*>(s + TILE_BYTES + soff[i]) = rb[i];
    }
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment offsets: rows i * 16 + frow share one swizzle key; the eight row blocks are immediates (i * 2048)
  const int fa0 = lds_off(wr * 128 + frow, fq), fa1 = lds_off(wr * 128 + frow, 4 + fq);
  const int fb0 = TILE_BYTES + lds_off(wc * 128 + frow, fq), fb1 = TILE_BYTES + lds_off(wc * 128 + frow, 4 + fq);

  const int KT = K >> 6;
  issue(0);
  commit(0);
  __syncthreads();
#define SB() __builtin_amdgcn_sched_barrier(0)
  // software pipeline, pinned: the B fragments of a half are read while the previous half's last MFMA groups run, the A
  // fragment of row block i + 1 before the eight MFMAs of row block i
  half8 bf[2][8], afc, afn;
#pragma unroll
  for (int j = 0; j < 8; ++j) bf[0][j] = as_half8(*reinterpret_cast<const u32x4*>(smem + fb0 + j * 2048));
  afc = as_half8(*reinterpret_cast<const u32x4*>(smem + fa0));
  for (int kt = 0; kt < KT; ++kt) {
    const char* s = smem + (kt & 1) * STAGE;
    const char* sn = smem + ((kt + 1) & 1) * STAGE;
    if (kt + 1 < KT) issue(kt + 1);
    SB();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // next A fragment: row block i + 1 of this half, or row block 0 of the next half (h = 1: of the next K-tile — read
        // after the barrier below, so it is fetched at the top of the next iteration instead)
        if (i < 7) afn = as_half8(*reinterpret_cast<const u32x4*>(s + (h ? fa1 : fa0) + (i + 1) * 2048));
        else if (h == 0) afn = as_half8(*reinterpret_cast<const u32x4*>(s + fa1));
        // the other half's B fragments trickle in one per row block
        if (h == 0) bf[1][i] = as_half8(*reinterpret_cast<const u32x4*>(s + fb1 + i * 2048));
        SB();
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = VN_MFMA_16x16x32(bf[h][j], afc, acc[i][j], 0, 0, 0);
        SB();
        afc = afn;
      }
    }
    // (spreading these 16 ds_write_b128 over the MFMA groups of the second half — two per group — is SLOWER: 816-875 vs
    //  873-972 TF/s at 4096^3.  With one wave per SIMD a 13-cycle store between two 16-cycle MFMAs delays the next MFMA.)
    if (kt + 1 < KT) commit((kt + 1) & 1);
    __syncthreads();
    if (kt + 1 < KT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bf[0][j] = as_half8(*reinterpret_cast<const u32x4*>(sn + fb0 + j * 2048));
      afc = as_half8(*reinterpret_cast<const u32x4*>(sn + fa0));
    }
    SB();
  }
#undef SB
  // epilogue (lab): a lane owns 4 consecutive n of one m per fragment
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int m = m0 + wr * 128 + i * 16 + frow, n = n0 + wc * 128 + j * 16 + 4 * fq;
      half4 o = {(half_t)acc[i][j][0], (half_t)acc[i][j][1], (half_t)acc[i][j][2], (half_t)acc[i][j][3]};
      *reinterpret_cast<half4*>(C + (long long)m * N + n) = o;
    }
}
}  // namespace

extern "C" int lab_gemm4w(const void* A, const void* B, void* C, int M, int N, int K, void* stream) {
  if (M % 256 || N % 256 || K % 64) return -1;
  const int tiles_n = N / 256;
  hipLaunchKernelGGL(gemm4w_kernel, dim3((M / 256) * tiles_n), dim3(256), 0, (hipStream_t)stream, (const half_t*)A,
                     (const half_t*)B, (half_t*)C, M, N, K, tiles_n);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
