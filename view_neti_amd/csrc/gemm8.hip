// 256x256 "8-phase" ping-pong tile of the GEMM / implicit-GEMM convolution family (gfx950 / CDNA4).
//
//   C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T )      same contract, argument block and epilogues as gemm_conv.hip
//
// Carries the same launches as the generic 256x256 tile (tile_hint 5) — the large convolutions of the VAE encoder and
// the UNet's 64x64 level, the CLIP projections; reference: the diffusers modules driven from training/coach.py:165-169,
// 197-198 and the to_q/to_k/to_v/to_out calls of models/xti_attention_processor.py:30-55 — with a different main loop
// (the structure /opt/skills/guides/cdna_hip_programming.md calls the 256^2 8-phase template):
//
//   * 8 waves (2 x 4), two per SIMD.  A wave owns four 64x32 quadrants of C: rows {h*128 + wr*64 .. +64}, columns
//     {j*128 + wc*32 .. +32}, h, j in {0,1} — so every wave reads every "half tile" (A0, A1 = rows 0..127 / 128..255 of
//     the block's A panel, B0, B1 likewise), and a half tile is the unit of staging and of register loading.
//   * LDS: 2 K-tile buffers x 4 half tiles x 16 KiB = 128 KiB, 128-byte rows with the 16-byte chunks XOR-swizzled
//     (conflict-free ds_read_b128), filled by LDS-DMA (buffer_load ... lds; swizzle on the source side, out-of-range
//     offsets return zeros = conv padding and M/N/K tails).
//   * a K-tile (BK = 64) is 4 phases, one quadrant of 16 x v_mfma_f32_16x16x32_f16 each:
//         P0: read A0(t)   [8 ds_read_b128]   stage A1(t+1)   C00 += A0.B0
//         P1: read B1(t)   [4]                stage B0(t+2)   C01 += A0.B1
//         P2: read A1(t)   [8]                stage A0(t+2)   C10 += A1.B0
//         P3: read B0(t+1) [4]                stage B1(t+2)   C11 += A1.B1
//     each phase = { ds_reads, 2 LDS-DMAs, s_waitcnt vmcnt(10), s_barrier, lgkmcnt(0), 16 MFMA at raised priority,
//     s_barrier }.  The two wave rows (wr = 0 / 1) run one barrier apart, so on every SIMD one wave issues its reads and
//     DMAs while its partner runs its MFMAs.
//   * a half tile is staged six phases before it is read and five stagings stay in flight across every wait
//     (vmcnt never 0 in the loop): position 7 + p of the staging stream is issued in phase p, the wait of phase p
//     retires position p + 2, which phase p + 1 reads.  RAW: wait in phase p (before its first barrier), read in
//     phase p + 1.  WAR: a slot is restaged two phases after the phase that issued its last reads (the staggered wave
//     row retires them one barrier later than the other).
//   * epilogue: the gemm_conv.hip one (C tile through LDS, bias / row-add / residual / activation / gate / GEGLU /
//     GroupNorm sums, EPI levels), split-K partials straight from the accumulators.
#include <type_traits>

#include "common.h"
#include "gemm_args.h"

namespace {

constexpr int BM = 256, NT = 512;
constexpr int HALF_BYTES = 128 * 128;  // an A half tile: 128 rows x 64 halfs
constexpr int GN_IMG = 5;

// (the lab switches of earlier rounds — per-section s_memtime stamps, removal experiments — live in
// tools/lab/attic/lab_switches.patch; the product translation unit carries none)
#define VN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define VN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// BN = 256: the structure of the header.  BN = 128 (the N = 128 convolutions of the VAE at 512^2 / 256^2 and the N = 320 / 640
// layers of the UNet, whose grids the 256-wide tile fills badly): same waves and A half tiles, B half tiles of 64 columns
// (a wave owns 16 columns of each), THREE K-tile buffers of 48 KiB, and two phases of 16 MFMAs per K-tile:
//     Q0: read A0(t) B0(t) B1(t) [12 ds_read_b128]   stage A0 B0 B1 of tile t+2 (4 LDS-DMAs)   C00 += A0.B0, C01 += A0.B1   vmcnt(10)
//     Q1: read A1(t)             [8]                  stage A1 of tile t+2       (2 LDS-DMAs)   C10 += A1.B0, C11 += A1.B1   vmcnt(8)
// (stream position 12 + 6t .. is issued in tile t; a half tile is read four phases after its staging and restaged two
// phases after its last read.)
//
// HALO (BN = 128, stride-1 3x3 forward convolutions with chunk-major K): the implicit GEMM fetches every input pixel nine
// times, and tools/lab/gemm8_parts.py shows the N = 128 convolutions of the VAE waiting for exactly that — staging alone
// (no MFMAs, no fragment reads) takes 60 % of the launch, whether or not the gather hits L2: 48 KiB per K-tile through a
// fill path that moves (bytes in flight) / (latency).  Here a block owns a 16 x 16 pixel tile instead of 256 consecutive
// pixels; the 18 x 18 x 64-channel input patch of a channel chunk is DMA'd into LDS ONCE (41 KiB, two chunk slots, the
// next chunk's patch trickles in one DMA per K-tile) and the A fragments of all nine taps are read straight out of it
// (pixel (y + dy, x + dx): a different LDS address, not a different copy).  Only the 16 KiB B tile is staged per K-tile
// (ring of three): 2.3x fewer bytes through the fill path.  Logical block row r <-> pixel (r / 16, r % 16) of the tile;
// the epilogue writes row r to that pixel, so results are bit-identical to the row-major tiles.
template <int BN, int EPI, bool CONV, bool HALO = false>
__global__ __launch_bounds__(NT) void gemm8_kernel(GemmArgs g) {
  static_assert(BN == 256 || BN == 128, "tile width");
  static_assert(!HALO || (BN == 128 && CONV), "the halo-patch loop exists for the 256x128 convolution tile");
  constexpr int PATCH_STRIDE = 41 * 1024;  // 324 pixels x 128 B = 40.5 KiB, rounded up to the 41 wave-DMAs that fill it
  constexpr bool H42 = HALO && BN == 128;  // halo, 256 x 128: the waves form a 4 x 2 grid (below)
  constexpr int NJB = BN / 128;                 // 16-column blocks a wave owns in each B half
  constexpr int HB_BYTES = (BN / 2) * 128;      // a B half tile
  constexpr int HB_DMA = BN / 128;              // LDS-DMAs per thread and B half tile (64 rows each)
  constexpr int NBUF = BN == 256 ? 2 : 3;
  constexpr int BUF_BYTES = 2 * HALF_BYTES + 2 * HB_BYTES;  // A0 A1 B0 B1 of one K-tile
  constexpr int CS_LD = BN + 8;
  constexpr int CS_BYTES = BM * CS_LD * 2;
  constexpr int LOOP_BYTES = HALO ? 2 * PATCH_STRIDE + 3 * 2 * HB_BYTES : NBUF * BUF_BYTES;
  constexpr int LDS_BYTES = CS_BYTES > LOOP_BYTES ? CS_BYTES : LOOP_BYTES;
  constexpr int GN_NG = BN / 4 + 2;
  const half_t* const e_gate = EPI == 2 ? g.gate_src : nullptr;
  half_t* const e_C2 = EPI == 2 ? g.C2 : nullptr;
  const int e_geglu = EPI == 2 ? g.geglu : 0;
  const int e_act = EPI == 2 ? g.act : 0;
  float* const e_gn_sums = EPI >= 1 ? g.gn_sums : nullptr;
  const half_t* const e_rowadd = EPI >= 1 ? g.rowadd : nullptr;
  const int e_conv = CONV ? g.conv_mode : 0;
  constexpr int GN_BYTES = EPI == 0 ? 0 : GN_IMG * GN_NG * 4 * 8;  // [image][group][S1.hi S1.lo S2.hi S2.lo] 64-bit words
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + GN_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware tile mapping (bijective for any tile count): consecutive ids on one XCD sweep N for a fixed M panel
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
  const int kz = blockIdx.z;
  // HALO: tile_m counts 16 x 16 pixel tiles in (image, tile row, tile column) order; row0 = the tile's first output row
  int hb = 0, hty = 0, htx = 0, row0 = m0;
  if constexpr (HALO) {
    const int tpr = g.Wo >> 4, tpi = (g.Ho >> 4) * tpr;
    hb = tile_m / tpi;
    const int rem = tile_m - hb * tpi;
    hty = rem / tpr;
    htx = rem - hty * tpr;
    row0 = (hb * g.Ho + hty * 16) * g.Wo + htx * 16;
  }
  auto rowmem = [&](const int r) { return HALO ? row0 + (r >> 4) * g.Wo + (r & 15) : m0 + r; };

  const half_t* Ab = g.A + (long long)bz * g.strideA;
  const half_t* Bb = g.B + (long long)bz * g.strideB;
  const __amdgpu_buffer_rsrc_t rsA = vn_make_rsrc(Ab, g.a_bytes);
  const __amdgpu_buffer_rsrc_t rsB = vn_make_rsrc(Bb, g.b_bytes);

  // ---- staging geometry: one LDS-DMA of the block covers 64 rows x 128 B; a half tile is two of them (j = 0, 1).
  // Lane -> (row lrow = tid / 8 of the 64, 16-byte slot tid % 8); the slot is lane-linear in LDS, so the swizzle
  // (chunk ^ ((row >> 1) & 7)) is applied to the source chunk.  Row slot r = 2 * h + j is block row h*128 + j*64 + lrow.
  //
  // Source offset of row slot r for a K-tile = a_base[r] + (a wave-uniform byte offset of the tile), valid where bit
  // `tap` of a_mask[r] is set, else VN_OOB (zeros):
  //   plain GEMM      a_base = row * lda * 2 + chunk,  tile offset = k0 * 2,  mask = 1 for rows < M
  //   conv, gather    a_base = offset of input pixel (oy*stride - pad, ox*stride - pad) of the row's output pixel,
  //                   tile offset = +(dy * Wi + dx) * pixel stride + channel offset, mask bit (3*dy + dx) = that tap is
  //                   inside the image (the zero padding)
  //   conv, dgrad s1  a_base = offset of pixel (oy + pad, ox + pad), tile offset = -(dy * Wi + dx) * ... (transposed
  //                   gather with unflipped taps), mask likewise
  // so the per-tile address work is scalar, plus an and / compare / add / select per row — no tap-boundary recompute.
  // (Fused upsampling and the stride-2 transposed gather are not linear in the tap: those launches stay on tile 5.)
  const int lrow = tid >> 3;
  const int gchunk = (tid & 7) ^ ((lrow >> 1) & 7);
  const float rcp_hw = CONV ? 1.0f / (float)(g.Ho * g.Wo) : 0.f, rcp_wo = CONV ? 1.0f / (float)g.Wo : 0.f;
  int a_base[4];
  uint32_t a_mask[4];
#pragma unroll
  for (int r = 0; r < (HALO ? 0 : 4); ++r) {
    const int m = m0 + (r >> 1) * 128 + (r & 1) * 64 + lrow;
    const bool ok = m < g.M;
    if constexpr (!CONV) {
      a_base[r] = (int)((long long)m * g.lda * 2) + gchunk * 16;
      a_mask[r] = ok ? 1u : 0u;
    } else {
      // (b, oy, ox) of output pixel m without integer division: m < 2^24 (checked by the launcher), so the float
      // quotient is off by at most one
      const int hw = g.Ho * g.Wo;
      int b = (int)((float)m * rcp_hw);
      int rem = m - b * hw;
      if (rem < 0) { b -= 1; rem += hw; } else if (rem >= hw) { b += 1; rem -= hw; }
      int oy = (int)((float)rem * rcp_wo);
      int ox = rem - oy * g.Wo;
      if (ox < 0) { oy -= 1; ox += g.Wo; } else if (ox >= g.Wo) { oy += 1; ox -= g.Wo; }
      const int py = e_conv == 1 ? oy * g.stride - g.pad_t : oy + g.pad_t;
      const int px = e_conv == 1 ? ox * g.stride - g.pad_l : ox + g.pad_l;
      a_base[r] = ((b * g.Hi + py) * g.Wi + px) * g.ldx2 + gchunk * 16;
      // tap (dy, dx) reads input pixel (py +- dy, px +- dx): three row bits x three column bits
      const int sgn = e_conv == 1 ? 1 : -1;
      uint32_t colbits = 0, mask = 0;
#pragma unroll
      for (int d = 0; d < 3; ++d) colbits |= ((unsigned)(px + sgn * d) < (unsigned)g.Wi ? 1u : 0u) << d;
#pragma unroll
      for (int d = 0; d < 3; ++d) mask |= ((unsigned)(py + sgn * d) < (unsigned)g.Hi ? colbits : 0u) << (3 * d);
      a_mask[r] = ok ? mask : 0u;
    }
  }
  uint32_t b_base[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + (r >> 1) * (BN / 2) + (r & 1) * 64 + lrow;  // (slots with r & 1 are unused when BN = 128)
    b_base[r] = (n < g.N) ? (uint32_t)((long long)n * g.ldb * 2) + gchunk * 16 : VN_OOB;  // + k0 * 2 stays out of range
  }

  const int nk_total = g.K / 64;
  const int kt_begin = kz * g.kt_per_split;
  const int kt_end = min(nk_total, kt_begin + g.kt_per_split);
  const int T = kt_end - kt_begin;

  // ---- the four staging streams (A0, A1, B0, B1): the next K-tile each will fetch (all wave-uniform) ----
  int a_kt[2] = {kt_begin, kt_begin}, b_kt[2] = {kt_begin, kt_begin};
  int a_tap[2], a_ci0[2];  // conv: tap and first channel of the stream's next tile (tap-major K order)
  {
    int tap = 0, ci0 = 0;
    if constexpr (CONV) {
      if (g.korder) {
        ci0 = (kt_begin / 9) * 64;
        tap = kt_begin - (kt_begin / 9) * 9;
      } else {
        tap = (kt_begin * 64) / g.Ci;
        ci0 = kt_begin * 64 - tap * g.Ci;
      }
    }
    a_tap[0] = a_tap[1] = tap;
    a_ci0[0] = a_ci0[1] = ci0;
  }
  // A staging is split in two: prep*() computes the two source offsets of the stream's next K-tile (and advances the
  // stream) — it runs inside the PREVIOUS phase's MFMA block, where VALU / SALU issue beside the matrix pipe for free —
  // and issue*() is just the two LDS-DMAs, so a phase's load section stays shorter than its partner's 16 MFMAs.
  const int tap_step = (e_conv == 2 ? -1 : 1) * g.ldx2;
  uint32_t nxt[4];  // prepared source offsets: [0..1] an A half, [2..3] B halves (BN = 128: B0 and B1 of one K-tile)
  auto prepA = [&](const int h) {
    const bool live = a_kt[h] < kt_end;  // stagings past the end keep the vmcnt bookkeeping uniform and fetch nothing
    int soff;
    uint32_t tapbit;
    if constexpr (!CONV) {
      soff = a_kt[h] * 128;
      tapbit = live ? 1u : 0u;
    } else {
      const int tap = a_tap[h];
      const int dy = (tap * 11) >> 5;  // tap / 3 for tap < 9
      const int dx = tap - 3 * dy;
      soff = (dy * g.Wi + dx) * tap_step + a_ci0[h] * 2;
      tapbit = live ? (1u << tap) : 0u;
      if (g.korder) {  // chunk-major K: the nine taps of a 64-channel chunk are consecutive K-tiles (their re-reads of the
        a_tap[h] += 1;  // same few input rows hit L2 instead of the fabric); costs nothing here, the offsets are linear
        if (a_tap[h] == 9) {
          a_tap[h] = 0;
          a_ci0[h] += 64;
        }
      } else {
        a_ci0[h] += 64;
        if (a_ci0[h] == g.Ci) {
          a_ci0[h] = 0;
          a_tap[h] += 1;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      nxt[j] = (a_mask[2 * h + j] & tapbit) ? (uint32_t)(a_base[2 * h + j] + soff) : VN_OOB;
    }
    a_kt[h] += 1;
  };
  auto prepB = [&](const int h) {  // BN = 256: into nxt[0..1]; BN = 128: into nxt[2 + h]
    const bool live = b_kt[h] < kt_end;
    const uint32_t soff = (uint32_t)b_kt[h] * 128u, dead = live ? 0u : VN_OOB;  // | VN_OOB: beyond any buffer (< 2 GiB)
    if constexpr (BN == 256) {
#pragma unroll
      for (int j = 0; j < 2; ++j) nxt[j] = (b_base[2 * h + j] + soff) | dead;
    } else {
      nxt[2 + h] = (b_base[2 * h] + soff) | dead;
    }
    b_kt[h] += 1;
  };
  auto issueA = [&](const int h, const int buf) {
    char* dst = smem + buf * BUF_BYTES + h * HALF_BYTES + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dma16(rsA, dst + j * 8192, nxt[j]);
  };
  auto issueB = [&](const int h, const int buf) {
    char* dst = smem + buf * BUF_BYTES + 2 * HALF_BYTES + h * HB_BYTES + wave * 1024;
    if constexpr (BN == 256) {
#pragma unroll
      for (int j = 0; j < 2; ++j) dma16(rsB, dst + j * 8192, nxt[j]);
    } else {
      dma16(rsB, dst, nxt[2 + h]);
    }
  };

  // ---- fragment reads: lane (frow = row of the 16-row block, fq = its 8-wide k chunk inside a k32 sub-step) ----
  const int frow = lane & 15;
  const int fq = lane >> 4;
  const int fkey = (frow >> 1) & 7;
  // byte offsets (buffer 0) of this lane's chunk for k sub-step 0 / 1; the K-tile buffer is toggled with ^ BUF_BYTES
  int rdA[2], rdB[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int ch = ((s * 4 + fq) ^ fkey) << 4;
    rdA[s] = (wr * 64 + frow) * 128 + ch;
    rdB[s] = 2 * HALF_BYTES + (wc * (16 * NJB) + frow) * 128 + ch;
  }
  half8 af[4][2], bf0[NJB][2], bf1[NJB][2];
  auto readA = [&](const int h) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        af[i][s] = as_half8(*reinterpret_cast<const u32x4*>(smem + rdA[s] + h * HALF_BYTES + i * 2048));
  };
  auto readB0 = [&](const int flip) {  // flip = BUF_BYTES: from the other K-tile buffer (BN = 256)
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        bf0[jb][s] = as_half8(*reinterpret_cast<const u32x4*>(smem + (rdB[s] ^ flip) + jb * 2048));
  };
  auto readB1 = [&]() {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        bf1[jb][s] = as_half8(*reinterpret_cast<const u32x4*>(smem + rdB[s] + HB_BYTES + jb * 2048));
  };

  // the bias of this lane's 16 columns, requested before the main loop (its L2 round trip would otherwise sit between
  // the last MFMA and the first C-tile write); out-of-range columns and a null bias read as zeros
  // HALO, BN = 128: the waves form a 4 x 2 grid of 64 x 64 tiles (wave tile 128 x 32 re-reads A from LDS four times per block and
  // makes the 256 x 128 tile LDS-read-bound: 160 KiB per K-tile against ~183 B/clk; 64 x 64 needs 128 KiB) — accumulator
  // [h][j][i][0] is then row block i of the wave's 64 rows, column block 2h + j of its 64 columns.
  const int wr4 = wave >> 1, wc2 = wave & 1;
  auto acc_row = [&](const int h, const int i) { return H42 ? wr4 * 64 + i * 16 + frow : h * 128 + wr * 64 + i * 16 + frow; };
  auto acc_col = [&](const int h, const int j, const int jb) {
    return H42 ? wc2 * 64 + (2 * h + j) * 16 + 4 * fq : j * (BN / 2) + wc * (16 * NJB) + jb * 16 + 4 * fq;
  };
  f32x4 bv[H42 ? 2 : 1][2][NJB];
  {
    const __amdgpu_buffer_rsrc_t rsBias = vn_make_rsrc(g.bias, g.bias ? (uint32_t)g.N * 4u : 0u);
#pragma unroll
    for (int h = 0; h < (H42 ? 2 : 1); ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
          bv[h][j][jb] = __builtin_bit_cast(f32x4, vn_buf_load16(rsBias, (uint32_t)(n0 + acc_col(h, j, jb)) * 4u));
  }
  f32x4 acc[2][2][4][NJB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) acc[h][j][i][jb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one quadrant: 16 MFMAs; operands swapped (D[row = n][col = m]) so a lane owns 4 consecutive n of one m.  PREP = the
  // address work of the NEXT phase's staging, free to be scheduled between the MFMAs.
#define VN_MMA(H, J, BF, PREP)                                                                                 \
  do {                                                                                                         \
    VN_WAIT_LGKM0();                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    PREP;                                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < 4; ++i)                \
        _Pragma("unroll") for (int jb = 0; jb < NJB; ++jb) acc[H][J][i][jb] =                                  \
            VN_MFMA_16x16x32(BF[jb][s], af[i][s], acc[H][J][i][jb], 0, 0, 0);                                  \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)
#define VN_PHASE_SYNC()                    \
  do {                                     \
    VN_WAIT_VM(10);                        \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
#define VN_PHASE_END()                     \
  do {                                     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

  // two quadrants that share the A fragments: 16 MFMAs (one 16-column block per B half)
#define VN_MMA2(H, PREP)                                                                                       \
  do {                                                                                                         \
    VN_WAIT_LGKM0();                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    PREP;                                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < 4; ++i) {              \
      acc[H][0][i][0] = VN_MFMA_16x16x32(bf0[0][s], af[i][s], acc[H][0][i][0], 0, 0, 0);                       \
      acc[H][1][i][0] = VN_MFMA_16x16x32(bf1[0][s], af[i][s], acc[H][1][i][0], 0, 0, 0);                       \
    }                                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)
#define VN_SYNC(n)                         \
  do {                                     \
    VN_WAIT_VM(n);                         \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
  if constexpr (HALO) {
    constexpr int BRING = 2 * PATCH_STRIDE;   // the B tiles live behind the two patch slots
    // Swizzle key of patch pixel column px (0..17): physical chunk = logical chunk ^ key.  A ds_read_b128 is served in lane
    // groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): eight lanes of one k-chunk and eight of its XOR-1 neighbour,
    // i.e. pixels dx + {0-3, 12-15} with chunk c and dx + {4-11} with chunk c ^ 1.  The row-major tiles' key (px >> 1) & 7
    // is conflict free only for dx = 0: taps with dx = 1, 2 (six of nine) hit two bank groups twice (rocprofv3 in round 4:
    // SQ_LDS_BANK_CONFLICT = 25 % of SQ_LDS_IDX_ACTIVE on this kernel, 0 on the row-major 256 x 256 tile).  With x_k = key of
    // pixels 2k, 2k + 1 the two window conditions { x0 x1 x2' x3' x4' x5' x6 x7 } and { x1 x2 x3' x4' x5' x6' x7 x8 }
    // (x' = x ^ 1) must both be permutations of 0..7, which forces x6 = x2, x8 = x0; the table below is one solution
    // (checked exhaustively over taps, k32 sub-steps and lane groups: tools/lab/patch_swizzle.py).  Removes a quarter of the
    // kernel's LDS cycles; its duration did not move (the loop does not wait for LDS bandwidth).
    auto patch_key = [](const int px) { return (int)((0x270745032ull >> (4 * (px >> 1))) & 7); };
    constexpr int PROW = 18 * 128;            // bytes per patch row (18 pixels x 64 channels)
    const int c_begin = kt_begin / 9, c_end = kt_end / 9;  // this split's channel chunks (the launcher aligns splits to chunks)
    // ---- patch staging: wave-DMA number d = 8j + wave (d < 41) covers the 64 consecutive 16-byte slots q = 64d + lane of
    // a patch slot; slot q = pixel p = q / 8 (row-major in the 18 x 18 patch), physical chunk q % 8, which holds the logical
    // chunk (q % 8) ^ ((px >> 1) & 7) — a 16-lane fragment read walks 16 consecutive px of one patch row, so keying the
    // swizzle by px keeps it conflict free for every tap.  Pixels outside the image (the zero padding) and the tail of
    // DMA 40 fetch nothing (out-of-range offset => zeros).  Wave 0 issues six DMAs per chunk, the others five.
    uint32_t pa_off[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = (j * 8 + wave) * 64 + lane, p = q >> 3, cs = q & 7;
      const int py = (p * 3641) >> 16;  // p / 18 for p < 3000
      const int px = p - py * 18;
      const int iy = hty * 16 + py - 1, ix = htx * 16 + px - 1;
      const bool ok = p < 324 && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      pa_off[j] = ok ? (uint32_t)(((hb * g.Hi + iy) * g.Wi + ix) * g.ldx2 + ((cs ^ patch_key(px)) << 4)) : VN_OOB;
    }
    const int np_wave = wave == 0 ? 6 : 5;  // patch DMAs of this wave per chunk
    auto issueP = [&](const int j, const int chunk) {  // + chunk * 128 keeps an out-of-range offset out of range (< 2 GiB)
      dma16(rsA, smem + (chunk & 1) * PATCH_STRIDE + (j * 8 + wave) * 1024, pa_off[j] + (uint32_t)chunk * 128u);
    };
    // tap (dy, dx) of output pixel (y, x) reads patch pixel (y + dy, x + dx) in the forward gather and (y + 2 - dy,
    // x + 2 - dx) in the stride-1 transposed one (the patch starts one pixel up and left of the tile)
    auto tap_dy = [&](const int tap) { const int dy = (tap * 11) >> 5; return e_conv == 2 ? 2 - dy : dy; };
    auto tap_dx = [&](const int tap) { const int dx = tap - 3 * ((tap * 11) >> 5); return e_conv == 2 ? 2 - dx : dx; };
    {
      constexpr int BBUF = 2 * HB_BYTES;  // 16 KiB per K-tile, ring of three
      auto issueBt = [&](const int kt, const int buf) {  // both 64-column halves of K-tile kt
        const uint32_t soff = (uint32_t)(kt_begin + kt) * 128u, dead = kt < T ? 0u : VN_OOB;
        char* dst = smem + BRING + buf * BBUF + wave * 1024;
        dma16(rsB, dst, (b_base[0] + soff) | dead);
        dma16(rsB, dst + HB_BYTES, (b_base[2] + soff) | dead);
      };
      // fragment reads: lane (frow = pixel column of the tile row, fq = 8-channel chunk of the k32 sub-step)
      int lb[3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) lb[dx] = (wr4 * 4) * PROW + (frow + dx) * 128 + ((fq ^ patch_key(frow + dx)) << 4);
      const int rb0 = (wc2 * 64 + frow) * 128 + ((fq ^ fkey) << 4);  // (sub-step 1: ^ 64 = chunk + 4)
      half8 bfh[4][2];  // the wave's four 16-column blocks of a B tile: read in Q0, held over both phases
      auto readAh = [&](const int half, const int ab) {  // tile rows 2 * half, 2 * half + 1 of the wave's four, at tap offset ab
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i) {
          af[i][0] = as_half8(*reinterpret_cast<const u32x4*>(smem + ab + i * PROW));
          af[i][1] = as_half8(*reinterpret_cast<const u32x4*>(smem + (ab ^ 64) + i * PROW));
        }
      };
      // 16 MFMAs: two row blocks x four column blocks x two k32 sub-steps (accumulator [h][j][i] = column block 2h + j).
      // PREP = scalar / address work of the NEXT K-tile, placed inside the MFMA block where SALU / VALU issue beside the
      // matrix pipe for free (as the row-major loops do with their staging offsets).
#define VN_MMA4(HALF, PREP)                                                                                    \
  do {                                                                                                         \
    VN_WAIT_LGKM0();                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    PREP;                                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 2 * HALF; i < 2 * HALF + 2; ++i) \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                     \
      acc[cb >> 1][cb & 1][i][0] =                                                                             \
          VN_MFMA_16x16x32(bfh[cb][s], af[i][s], acc[cb >> 1][cb & 1][i][0], 0, 0, 0);                         \
    }                                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)
      // The loop is UNROLLED OVER THE NINE TAPS of a channel chunk, so that everything a K-tile's load sections need besides
      // the loads themselves is an immediate or a register prepared outside the tile: tap (dy, dx) -> patch offset, B ring
      // slots (t % 3 == tap % 3 because 9 % 3 == 0), the patch piece tap - 1.  Round 4: with that arithmetic in the loop — tap
      // decode, two select chains over registers, slot counters: ~70 instructions of issue at the head of every Q0 load
      // section against the partner's 258 cycles of MFMAs — a barrier-to-barrier interval took ~450 cycles (stamps; SQ
      // counters: MFMA busy 40 % of the kernel, the row-major 256 x 256 tile 59 %).  The row-major loops had shown the same
      // effect (+21 %) with their address work in the load section; hoisting it into the MFMA block as they do does not work
      // here — the selects compile to branches, which the scheduler cannot weave between MFMAs.
      //   abt[k]   LDS offset of the A fragments (tile row 0, k32 sub-step 0) for the k-th K-tile of a chunk; the transposed
      //            gather (dgrad) reads patch pixel (y + 2 - dy, x + 2 - dx) = the forward offset of tap 8 - k
      //   bo0/bo1  source offsets of the two halves of the B tile staged next (+128 B per K-tile)
      //   pp[j]    source offset of patch piece j of the NEXT chunk (+128 B per chunk)
      int abt[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int tp = e_conv == 2 ? 8 - k : k, dy = tp / 3, dx = tp - 3 * dy;
        abt[k] = (dx == 0 ? lb[0] : (dx == 1 ? lb[1] : lb[2])) + dy * PROW + (c_begin & 1) * PATCH_STRIDE;
      }
      uint32_t bo0 = b_base[0] + (uint32_t)(kt_begin + 2) * 128u, bo1 = b_base[2] + (uint32_t)(kt_begin + 2) * 128u;
      uint32_t pp[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) pp[j] = pa_off[j] + (uint32_t)(c_begin + 1) * 128u;
      const bool w0 = wave == 0;  // wave 0 stages six patch pieces per chunk, the others five
      char* const bring = smem + BRING + wave * 1024;
      auto readBh = [&](const int slot) {  // slot: compile-time after unrolling
        const char* b = smem + BRING + slot * BBUF;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          bfh[cb][0] = as_half8(*reinterpret_cast<const u32x4*>(b + cb * 2048 + rb0));
          bfh[cb][1] = as_half8(*reinterpret_cast<const u32x4*>(b + cb * 2048 + (rb0 ^ 64)));
        }
      };
      // ---- prologue: the whole patch of chunk 0, B tiles 0 and 1 ----
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j < np_wave) issueP(j, c_begin);
      issueBt(0, 0);
      issueBt(1, 1);
      VN_WAIT_VM(2);  // patch 0 and B tile 0 have landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first
      __builtin_amdgcn_sched_barrier(0);
      // K-tile (chunk c, tap): Q0 reads the wave's four column blocks of B(t) and two of its four tile rows at the tap's
      // offset [12 ds_read_b128], stages B(t + 2) [its buffer was last read in Q0(t - 1)] [+ one DMA of chunk c + 1's patch
      // at taps 1..6: its slot was last read in Q1 of the previous chunk's tap 8, three phases before]; Q1 reads the other
      // two tile rows [4] and waits for B(t + 1) (read in the next Q0): everything but this tile's stagings, which come
      // later in the stream.  Stagings past the end of K (the last chunk's taps 7, 8; patch pieces behind the last chunk)
      // are issued with out-of-range offsets: they fetch nothing and keep the counted waits uniform.
      // (Round 4, measured against this loop, all bit-identical: eight reads per phase with column blocks 2, 3 of B(t + 1)
      //  prefetched in Q1 and the wait moved into Q0: equal; every phase's reads issued behind the previous phase's MFMAs,
      //  interleaved with them: +5 %, strictly after them: +9 % slower.)
      auto ktile = [&](auto tap_c, const bool last_chunk, const int pslot) {
        constexpr int tap = decltype(tap_c)::value;
        constexpr int rslot = tap % 3, sslot = (tap + 2) % 3;
        const int ab = abt[tap];
        // Q0
        readBh(rslot);
        readAh(0, ab);
        if constexpr (tap >= 1 && tap <= 6) {
          if (tap < 6 || w0) {
            dma16(rsA, smem + pslot + ((tap - 1) * 8) * 1024 + wave * 1024, last_chunk ? VN_OOB : pp[tap - 1]);
          }
        }
        {
          const uint32_t dead = (last_chunk && tap >= 7) ? VN_OOB : 0u;
          dma16(rsB, bring + sslot * BBUF, bo0 | dead);
          dma16(rsB, bring + sslot * BBUF + HB_BYTES, bo1 | dead);
          bo0 += 128u;
          bo1 += 128u;
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        VN_MMA4(0, (void)0);
        VN_PHASE_END();
        // Q1
        readAh(1, ab);
        if constexpr (tap >= 1 && tap <= 5) {
          VN_SYNC(3);
        } else if constexpr (tap == 6) {
          if (w0) {
            VN_SYNC(3);
          } else {
            VN_SYNC(2);
          }
        } else {
          VN_SYNC(2);
        }
        VN_MMA4(1, (void)0);
        VN_PHASE_END();
      };
      for (int c = c_begin; c < c_end; ++c) {
        const bool last_chunk = c + 1 >= c_end;
        const int pslot = ((c + 1) & 1) * PATCH_STRIDE;  // where the next chunk's patch goes
        ktile(std::integral_constant<int, 0>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 1>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 2>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 3>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 4>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 5>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 6>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 7>{}, last_chunk, pslot);
        ktile(std::integral_constant<int, 8>{}, last_chunk, pslot);
        // the next chunk reads the other patch slot; its successor's pieces come from 64 channels further on
        const int flip = (c & 1) ? -PATCH_STRIDE : PATCH_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) abt[k] += flip;
#pragma unroll
        for (int j = 0; j < 6; ++j) pp[j] += 128u;
      }
#undef VN_MMA4
    }
    // (The same loop was built for the 256x256 tile — A half tiles out of the patch, B half tiles staged as in the header —
    // and measured within +-3 % of the row-major one on every convolution of the step: that tile moves 64 KiB per 2048
    // MFMA cycles and is not bound by the fill path.  Not kept.)
  } else if constexpr (BN == 256) {
    // ---- prologue: tile 0 and three half tiles of tile 1 (stream positions 0..6) ----
    prepB(0);
    issueB(0, 0);
    prepA(0);
    issueA(0, 0);
    prepB(1);
    issueB(1, 0);
    prepA(1);
    issueA(1, 0);
    prepB(0);
    issueB(0, 1);
    prepA(0);
    issueA(0, 1);
    prepB(1);
    issueB(1, 1);
    prepA(1);        // A1 of tile 1: issued in phase 0
    VN_WAIT_VM(10);  // positions 0, 1 (B0, A0 of tile 0) have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    readB0(0);
    VN_WAIT_LGKM0();  // retired before the stagger barrier: the slot is restaged in phase 1
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);

    int cur = 0;
    for (int t = 0; t < T; ++t) {
      // P0
      readA(0);
      issueA(1, cur ^ 1);
      VN_PHASE_SYNC();
      VN_MMA(0, 0, bf0, prepB(0));
      VN_PHASE_END();
      // P1
      readB1();
      issueB(0, cur);
      VN_PHASE_SYNC();
      VN_MMA(0, 1, bf1, prepA(0));
      VN_PHASE_END();
      // P2
      readA(1);
      issueA(0, cur);
      VN_PHASE_SYNC();
      VN_MMA(1, 0, bf0, prepB(1));
      VN_PHASE_END();
      // P3
      readB0(BUF_BYTES);
      issueB(1, cur);
      VN_PHASE_SYNC();
      VN_MMA(1, 1, bf1, prepA(1));
      VN_PHASE_END();
      cur ^= 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        rdA[s] ^= BUF_BYTES;
        rdB[s] ^= BUF_BYTES;
      }
    }
  } else {
    // ---- prologue: tiles 0 and 1 (stream positions 0..11: [A0 A0 B0 B1 | A1 A1] per tile) ----
#pragma unroll
    for (int t0 = 0; t0 < 2; ++t0) {
      prepA(0);
      prepB(0);
      prepB(1);
      issueA(0, t0);
      issueB(0, t0);
      issueB(1, t0);
      prepA(1);
      issueA(1, t0);
    }
    prepA(0);  // tile 2's A0 B0 B1: issued in Q0 of tile 0
    prepB(0);
    prepB(1);
    VN_WAIT_VM(8);  // positions 0..3 (A0, B0, B1 of tile 0) have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);
    int stg = 2;  // buffer of the tile being staged: (t + 2) % 3
    for (int t = 0; t < T; ++t) {
      // Q0
      readB0(0);
      readB1();
      readA(0);
      issueA(0, stg);
      issueB(0, stg);
      issueB(1, stg);
      VN_SYNC(10);
      VN_MMA2(0, prepA(1));
      VN_PHASE_END();
      // Q1
      readA(1);
      issueA(1, stg);
      VN_SYNC(8);
      VN_MMA2(1, (prepA(0), prepB(0), prepB(1)));
      VN_PHASE_END();
      stg = stg == 2 ? 0 : stg + 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // reads move on to the next buffer: (t + 1) % 3
        rdA[s] = rdA[s] >= 2 * BUF_BYTES ? rdA[s] - 2 * BUF_BYTES : rdA[s] + BUF_BYTES;
        rdB[s] = rdB[s] >= 2 * BUF_BYTES ? rdB[s] - 2 * BUF_BYTES : rdB[s] + BUF_BYTES;
      }
    }
#undef VN_MMA2
#undef VN_SYNC
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the (zero-filling) stagings past the end must have landed
  __syncthreads();                                                // before the epilogue reuses the buffers as its C tile
#undef VN_MMA
#undef VN_PHASE_SYNC
#undef VN_PHASE_END

  // acc[h][j][i][jb][e]  <->  block row h*128 + wr*64 + i*16 + frow, block column j*(BN/2) + wc*16*NJB + jb*16 + 4*fq + e
  // ---- split-K: raw f32 partials straight to the workspace ----
  if (g.ksplit > 1) {
    float* ws = g.ws + ((long long)(kz * g.batch + bz) * g.M) * g.N;
    const __amdgpu_buffer_rsrc_t rsW = vn_make_rsrc(ws, 0x7fffffffu);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jb = 0; jb < NJB; ++jb) {
            const int m = rowmem(acc_row(h, i));
            const int n = n0 + acc_col(h, j, jb);
            if (m < g.M && n < g.N) {
              float* p = ws + (long long)m * g.N + n;
              if (n + 4 <= g.N && (g.N & 3) == 0) {
                vn_st16_wt(rsW, (uint32_t)(((long long)m * g.N + n) * 4), acc[h][j][i][jb]);
              } else {
                for (int e = 0; e < 4 && n + e < g.N; ++e) p[e] = acc[h][j][i][jb][e];
              }
            }
          }
    return;
  }

  if (e_gn_sums && !(HALO && g.gn_hw == g.Ho * g.Wo)) {  // (the one-image-per-tile path of the halo tile does not use it)
    vn_u64* gacc = reinterpret_cast<vn_u64*>(smem + LDS_BYTES);
    for (int i = tid; i < GN_IMG * GN_NG * 4; i += NT) gacc[i] = 0;
  }
  // ---- epilogue phase 1: acc -> (alpha, bias, act) -> LDS tile Cs[BM][CS_LD] ----
  // Rows are 528 B apart (132 dwords = 4 mod 32 banks), so the 16 rows a ds_write_b64 lane group covers would hit every
  // bank pair twice; rows with bit 3 set therefore store the two 8-byte halves of each 16-byte chunk swapped (bank + 2),
  // which phase 2 undoes in registers (there the rows of a thread all share that bit: it is wave-uniform).
  {
    const int nsw = ((frow >> 3) & 1) << 2;  // element offset of the half swap
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int jb = 0; jb < NJB; ++jb) {
            const int ml = acc_row(h, i);
            const int nl = acc_col(h, j, jb) ^ nsw;
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              o[e] = (half_t)apply_act(acc[h][j][i][jb][e] * g.alpha + bv[H42 ? h : 0][j][jb][e], e_act);
            *reinterpret_cast<half4*>(smem + ((size_t)ml * CS_LD + nl) * 2) = o;
          }
  }
  __syncthreads();

  // ---- epilogue phase 2: coalesced row-major stores with the fused operands.  A thread owns one 8-column chunk
  // (column c = 8 * (tid % 32)) of rows tid / 32 + 16 * it, it = 0..15, handled U rows at a time: the C chunks (LDS) and
  // every fused operand (residual, row-add, gate: buffer loads, out-of-range => zeros) of a batch are requested before
  // any of them is used, so a tile pays 16 / U memory round trips instead of 16 serialised ones. ----
  half_t* Cb = reinterpret_cast<half_t*>(g.C) + (long long)bz * g.strideC;
  const __amdgpu_buffer_rsrc_t rsC = vn_make_rsrc(Cb, 0x7fffffffu);
  const __amdgpu_buffer_rsrc_t rsC2 = vn_make_rsrc(e_C2, e_C2 ? 0x7fffffffu : 0u);
  const half_t* Rb = reinterpret_cast<const half_t*>(g.resid);
  if (Rb) Rb += (long long)bz * g.strideC;
  constexpr int CPR = BN / 8;
  constexpr int U = EPI == 2 ? 2 : (EPI == 1 ? 4 : (BN == 256 ? 8 : 4));
  static_assert(NT % CPR == 0 && CPR <= 32 && (BM * CPR) % NT == 0, "a thread keeps one 8-column chunk over all its rows");
  vn_u64* gacc = reinterpret_cast<vn_u64*>(smem + LDS_BYTES);
  const bool gn = e_gn_sums != nullptr;
  const float rcp_gnhw = gn ? 1.0f / (float)g.gn_hw : 0.f, rcp_rpg = e_rowadd ? 1.0f / (float)g.rows_per_group : 0.f;
  const int gn_img0 = gn ? row0 / g.gn_hw : 0, gn_g0t = gn ? n0 / g.gn_cpg : 0;
  const int c = (tid % CPR) * 8;
  const int n = n0 + c;
  const int gn_glo = gn ? n / g.gn_cpg : 0;
  const int gn_split = gn ? (gn_glo + 1) * g.gn_cpg - n : 8;  // columns [0, split) of the chunk are in group lo
  int gn_img = -1;
  float s_lo = 0.f, q_lo = 0.f, s_hi = 0.f, q_hi = 0.f;
  // flush = wave-uniform: the lanes that share a chunk column (lane % CPR) are summed with cross-lane moves first, so a
  // wave issues CPR x 4 LDS atomics on mostly distinct addresses instead of 256 colliding ones
  auto gn_flush = [&]() {
    const int iref = __builtin_amdgcn_readfirstlane(gn_img);
    const bool uni = __all(gn_img == iref) && iref >= 0;
    if (uni) {
#pragma unroll
      for (int off = 32; off >= CPR; off >>= 1) {
        s_lo += __shfl_xor(s_lo, off);
        q_lo += __shfl_xor(q_lo, off);
        s_hi += __shfl_xor(s_hi, off);
        q_hi += __shfl_xor(q_hi, off);
      }
    }
    if (gn_img >= 0 && (!uni || lane < CPR)) {
      // integer (fixed-point) atomics: the totals do not depend on the order the lanes / waves / blocks arrive in
      vn_u64* a = gacc + ((gn_img - gn_img0) * GN_NG + (gn_glo - gn_g0t)) * 4;
      vn_fx_add2(a, s_lo, q_lo);
      if (gn_split < 8) vn_fx_add2(a + 4, s_hi, q_hi);
    }
    s_lo = q_lo = s_hi = q_hi = 0.f;
  };
  const bool full_chunk = n + 8 <= g.N;  // false only in the last column chunk of an N that is no multiple of 8
  const bool swap_halves = ((tid / CPR) >> 3) & 1;  // bit 3 of this thread's rows (the same for all of them, and wave-uniform)
  const __amdgpu_buffer_rsrc_t rsR = vn_make_rsrc(Rb, Rb ? 0x7fffffffu : 0u);
  const __amdgpu_buffer_rsrc_t rsRA = vn_make_rsrc(e_rowadd, e_rowadd ? 0x7fffffffu : 0u);
  const __amdgpu_buffer_rsrc_t rsG = vn_make_rsrc(e_gate, e_gate ? 0x7fffffffu : 0u);
  const int r_first = tid / CPR;
  constexpr int RPP = NT / CPR;  // rows per pass of the block: 16 (BN = 256) or 32 (BN = 128)
  // A halo tile is 16 x 16 pixels of ONE image (hb): when the GroupNorm / row-add groups are whole images (they are for every
  // launch of the step: gn_hw = rows_per_group = Ho * Wo) the image of a row need not be derived per row — the per-row float
  // quotient, the wave vote and the conditional flush were a third of the instructions of this loop, and the time-embedding
  // chunk is one load per thread instead of sixteen (round 4: the fused statistics had made the VAE's 512^2 convolutions
  // 408 us where the plain epilogue takes 322).
  const bool gn_one = HALO && gn && g.gn_hw == g.Ho * g.Wo;
  const bool ra_one = HALO && EPI >= 1 && e_rowadd && g.rows_per_group == g.Ho * g.Wo;
  half8 av_tile = half8{0, 0, 0, 0, 0, 0, 0, 0};
  if (ra_one) av_tile = as_half8(vn_buf_load16(rsRA, full_chunk ? (uint32_t)(((long long)hb * g.ld_rowadd + n) * 2) : VN_OOB));
  if (gn_one) gn_img = hb;
  for (int it0 = 0; it0 < BM / RPP; it0 += U) {
    half8 cv[U], rv[U], av[U], gv[U], gv2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r_first + RPP * (it0 + u);
      const int m = rowmem(r);
      const bool ok = m < g.M && full_chunk;
      u32x4 t = *reinterpret_cast<const u32x4*>(smem + ((size_t)r * CS_LD + c) * 2);
      if (swap_halves) t = u32x4{t[2], t[3], t[0], t[1]};
      cv[u] = as_half8(t);
      if (Rb) rv[u] = as_half8(vn_buf_load16(rsR, ok ? (uint32_t)(((long long)m * g.ldr + n) * 2) : VN_OOB));
      if (EPI >= 1) {
        if (ra_one) {
          av[u] = av_tile;
        } else if (e_rowadd) {
          int grp = (int)((float)m * rcp_rpg);  // m / rows_per_group, m < 2^24: the float quotient is off by at most one
          const int rem = m - grp * g.rows_per_group;
          grp += rem >= g.rows_per_group ? 1 : (rem < 0 ? -1 : 0);
          av[u] = as_half8(vn_buf_load16(rsRA, ok ? (uint32_t)(((long long)grp * g.ld_rowadd + n) * 2) : VN_OOB));
        }
      }
      if (EPI == 2 && e_gate) {
        const long long go = (long long)m * g.ld_gate + (e_geglu == 2 ? 2 * n : n);
        gv[u] = as_half8(vn_buf_load16_once(rsG, ok ? (uint32_t)(go * 2) : VN_OOB));
        gv2[u] = as_half8(vn_buf_load16_once(rsG, (ok && e_geglu == 2) ? (uint32_t)(go * 2 + 16) : VN_OOB));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r_first + RPP * (it0 + u);
      const int m = rowmem(r);
      const bool valid = m < g.M && n < g.N;
      if (gn && !gn_one) {
        int img = gn_img;
        if (valid) {
          img = (int)((float)m * rcp_gnhw);
          const int rem = m - img * g.gn_hw;
          img += rem >= g.gn_hw ? 1 : (rem < 0 ? -1 : 0);
        }
        if (__any(img != gn_img)) {
          gn_flush();
          gn_img = img;
        }
      }
      if (!valid) continue;
      half8 v = cv[u];
      if (full_chunk) {
        if (EPI >= 1 && e_rowadd) {
          v = vn_add8(v, av[u]);
        }
        if (Rb) {
          v = vn_add8(v, rv[u]);
        }
        if (e_geglu == 2) {
          // GEGLU backward: v = d(h * gelu(g)) for 8 outputs; the saved pre-activation holds [h0..3 g0..3 h4..7 g4..7]
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const half8 pre = c2 == 0 ? gv[u] : gv2[u];
            half8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = (float)v[4 * c2 + e], hh = (float)pre[e], gg = (float)pre[4 + e];
              float cdf, xpdf;
              vn_gelu_parts(gg, cdf, xpdf);
              o[e] = (half_t)(d * gg * cdf);
              o[4 + e] = (half_t)(d * hh * (cdf + xpdf));
            }
            vn_st16_wt(rsC, (uint32_t)(((long long)m * g.ldc + 2 * n + 8 * c2) * 2), o);
          }
          continue;
        }
        if (e_gate) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * act_grad((float)gv[u][e], g.gate_act));
        }
        // (the GEGLU pre-activation p, and a pre-activation stored beside its activated copy C2, are read by the backward only)
        if (e_geglu == 1 || (e_C2 && e_geglu == 0)) vn_st16_wt_saved(rsC, (uint32_t)(((long long)m * g.ldc + n) * 2), v);
        else vn_st16_wt(rsC, (uint32_t)(((long long)m * g.ldc + n) * 2), v);
        if (e_geglu == 1) {
          half4 o2;
#pragma unroll
          for (int e = 0; e < 4; ++e) o2[e] = (half_t)((float)v[e] * vn_gelu_erf((float)v[4 + e]));
          *reinterpret_cast<half4*>(e_C2 + (long long)m * g.ldc2 + (n >> 1)) = o2;
        }
        if (gn) {
          {
            // two channels per v_dot2 (sum: against (1, 1); sum of squares: against itself): 8 VALU per row instead of 24 —
            // the statistics were 8-12 % of the launches that carry them.  A pair never straddles the group boundary: the
            // split is even because the channels per group are (vneti_gemm_f16 sends odd group sizes to the generic tiles).
            const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              const half2v p2 = {v[e], v[e + 1]};
              if (e < gn_split) {
                s_lo = VN_FDOT2(p2, one2, s_lo);
                q_lo = VN_FDOT2(p2, p2, q_lo);
              } else {
                s_hi = VN_FDOT2(p2, one2, s_hi);
                q_hi = VN_FDOT2(p2, p2, q_hi);
              }
            }
          }
        }
        if (e_C2 && e_geglu == 0) {
          half8 o2;
#pragma unroll
          for (int e = 0; e < 8; ++e) o2[e] = (half_t)apply_act((float)v[e], g.act2);
          vn_st16_wt(rsC2, (uint32_t)(((long long)m * g.ldc2 + n) * 2), o2);
        }
      } else {  // ragged last chunk (N % 8 != 0): element-wise, operands straight from memory
        const half_t* radd = e_rowadd ? e_rowadd + (long long)(m / g.rows_per_group) * g.ld_rowadd + n : nullptr;
        for (int e = 0; e < 8 && n + e < g.N; ++e) {
          float x = (float)v[e];
          if (radd) x = (float)(half_t)(x + (float)radd[e]);
          if (Rb) x = (float)(half_t)(x + (float)Rb[(long long)m * g.ldr + n + e]);
          if (e_gate) x = (float)(half_t)(x * act_grad((float)e_gate[(long long)m * g.ld_gate + n + e], g.gate_act));
          Cb[(long long)m * g.ldc + n + e] = (half_t)x;
          if (e_C2) e_C2[(long long)m * g.ldc2 + n + e] = (half_t)apply_act((float)(half_t)x, g.act2);
          if (gn) {
            const float xs = (float)(half_t)x;
            if (e < gn_split) {
              s_lo += xs;
              q_lo += xs * xs;
            } else {
              s_hi += xs;
              q_hi += xs * xs;
            }
          }
        }
      }
    }
  }
  if (gn_one) {
    // one image per tile: no LDS atomics — the 32 threads that share a chunk column are summed in a FIXED order (cross-lane
    // moves inside the wave, then the eight waves' partials through a 2 KiB table), so the totals are still run-to-run
    // identical; one fixed-point encode + four global atomics per (chunk column, group half)
    float v4[4] = {s_lo, q_lo, s_hi, q_hi};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v4[k] += __shfl_xor(v4[k], 32);
      v4[k] += __shfl_xor(v4[k], 16);
    }
    static_assert(CPR == 16 || !HALO, "the halo tile is 128 columns wide");
    float* tab = reinterpret_cast<float*>(smem + LDS_BYTES);  // [wave][chunk column][4], inside the gacc area (never zeroed here)
    if (lane < 16) *reinterpret_cast<f32x4*>(tab + (wave * 16 + lane) * 4) = f32x4{v4[0], v4[1], v4[2], v4[3]};
    __syncthreads();
    if (tid < 32) {
      const int col = tid >> 1, half = tid & 1;
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) {
        sv += tab[(w8 * 16 + col) * 4 + 2 * half];
        qv += tab[(w8 * 16 + col) * 4 + 2 * half + 1];
      }
      const int nc = n0 + col * 8;
      const int glo = nc / g.gn_cpg;
      const int split = (glo + 1) * g.gn_cpg - nc;
      if (nc < g.N && (half == 0 || split < 8)) {
        const int slot = tile_m % g.gn_slots;
        vn_u64* dst = reinterpret_cast<vn_u64*>(e_gn_sums) + (((long long)hb * g.gn_slots + slot) * g.gn_G + glo + half) * 4;
        vn_fx_add2(dst, sv, qv);
      }
    }
  } else if (gn) {
    gn_flush();
    __syncthreads();
    const int slot = tile_m % g.gn_slots;
    for (int i = tid; i < GN_IMG * GN_NG; i += NT) {
      const vn_u64* src = gacc + 4 * i;
      if ((src[0] | src[1] | src[2] | src[3]) == 0) continue;
      const int img = gn_img0 + i / GN_NG, grp = gn_g0t + i % GN_NG;
      vn_u64* dst = reinterpret_cast<vn_u64*>(e_gn_sums) + (((long long)img * g.gn_slots + slot) * g.gn_G + grp) * 4;
#pragma unroll
      for (int w = 0; w < 4; ++w) atomicAdd(dst + w, src[w]);
    }
  }
}

inline int epilogue_level8(const GemmArgs& g) {
  if (g.gate_src || g.C2 || g.geglu || g.act) return 2;
  return (g.gn_sums || g.rowadd) ? 1 : 0;
}

}  // namespace

// f16 output only; the caller (vneti_gemm_f16) has validated the descriptor, set ksplit / kt_per_split and launches the
// split-K reduce itself.  Returns VNETI_EUNSUP for what the requested tile does not carry (f32 output, fused upsample,
// stride-2 transposed gather; halo: anything but a stride-1 pad-1 3x3 convolution (forward or transposed gather) on a
// 16-pixel grid with chunk-major K; its split-K splits are rounded to whole channel chunks).
int vneti_launch_gemm8(void* gemm_args, int bn, int halo, hipStream_t st) {
  GemmArgs& g = *reinterpret_cast<GemmArgs*>(gemm_args);
  if (g.out_f32 || (g.conv_mode && (g.ups || (g.conv_mode == 2 && g.stride == 2))) ||
      (g.M >= (1 << 24) && (g.conv_mode || g.rowadd || g.gn_sums)))  // float-reciprocal row arithmetic: rows < 2^24
    return VNETI_EUNSUP;
  if (halo && (bn != 128 || g.conv_mode < 1 || g.conv_mode > 2 || g.stride != 1 || g.pad_t != 1 || g.pad_l != 1 || g.Hi != g.Ho ||
               g.Wi != g.Wo || (g.Ho & 15) || (g.Wo & 15) || (g.Ci & 63) || g.K != 9 * g.Ci || !g.korder ||
               g.batch != 1 || g.M != (g.M / (g.Ho * g.Wo)) * g.Ho * g.Wo))
    return VNETI_EUNSUP;
  if (halo && g.ksplit > 1) {  // splits own whole 64-channel chunks (nine K-tiles each): the patch logic stays per chunk
    const int nchunk = g.Ci / 64, cps = cdiv(nchunk, g.ksplit);
    g.kt_per_split = 9 * cps;
    g.ksplit = cdiv(nchunk, cps);
  }
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, bn);
  const dim3 grid(g.tiles_m * g.tiles_n, g.batch, g.ksplit), block(NT);
  const int epi = epilogue_level8(g);
#define VN_GO(E, C)                                                                        \
  do {                                                                                     \
    if (bn == 256) hipLaunchKernelGGL((gemm8_kernel<256, E, C>), grid, block, 0, st, g);   \
    else hipLaunchKernelGGL((gemm8_kernel<128, E, C>), grid, block, 0, st, g);             \
  } while (0)
#define VN_GO_HALO(E) hipLaunchKernelGGL((gemm8_kernel<128, E, true, true>), grid, block, 0, st, g)
  if (halo) {
    if (epi == 2) VN_GO_HALO(2); else if (epi == 1) VN_GO_HALO(1); else VN_GO_HALO(0);
  } else if (g.conv_mode) {
    if (epi == 2) VN_GO(2, true); else if (epi == 1) VN_GO(1, true); else VN_GO(0, true);
  } else {
    if (epi == 2) VN_GO(2, false); else if (epi == 1) VN_GO(1, false); else VN_GO(0, false);
  }
#undef VN_GO
#undef VN_GO_HALO
  return vneti_check_launch("gemm8_kernel");
}
