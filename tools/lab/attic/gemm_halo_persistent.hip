// LAB / NOT BUILT (round 4): the halo-patch convolution tile as its own kernel, PERSISTENT over tiles (the next tile's first
// patch and first two weight tiles are staged by the current tile's last chunk).  Bit-identical to the product tile and
// 2-10 % faster on isolated multi-round launches (512^2 128->128: 357 -> 322 us), but the train step was 1.7 % SLOWER with
// it on one box (38.6 vs 37.9-38.0 steps/s, three alternating runs): a persistent block per CU holds every CU for the whole
// launch, so the side-stream launches of the step (text path beside the VAE) no longer slip in between tile rounds.
// Kept for the record; the product tile is gemm8_kernel<128, EPI, true, true> in view_neti_amd/csrc/gemm8.hip.

// Halo-patch 3x3 convolution tile (tile_hint 18) of the GEMM / implicit-GEMM family, gfx950 / CDNA4.
//
//   C[M, N] = epilogue( A (x) W )   stride-1 pad-1 3x3 convolution, forward gather (conv_mode 1) or the transposed gather of
//   the input gradient (conv_mode 2), NHWC f16 image, chunk-major K (K-tile = (64-channel chunk, tap)); same argument block
//   and fused epilogues as gemm_conv.hip / gemm8.hip.
//
// Carries the step's stride-1 3x3 convolutions and their input gradients: ResnetBlock2D.conv1 / conv2 of the UNet and of the
// VAE encoder driven from training/coach.py:165-169 (vae.encode) and :197-198 (unet forward; the backward of :211).
//
// Structure (history in DESIGN.md section 4):
//   * a block of 8 waves owns a 16 x 16 pixel tile x 128 output channels.  The 18 x 18 x 64-channel input patch of a channel
//     chunk sits in LDS for all nine taps (40.5 KiB, two slots: the next chunk's patch trickles in one wave-DMA per K-tile);
//     tap (dy, dx) is an LDS address offset, not another copy.  Only the 16 KiB weight tile is staged per K-tile (ring of
//     three).  2.3x fewer bytes through the LDS fill path than the row-major 256 x 128 tile.
//   * waves as a 4 x 2 grid of 64 x 64 tiles, two per SIMD in ping-pong: the wave rows {0..3} / {4..7} run one s_barrier apart,
//     so in every barrier-to-barrier interval one wave of a SIMD issues its 16 MFMAs while its partner loads.
//   * the K loop is UNROLLED OVER THE NINE TAPS of a chunk: every address a load section needs is an immediate or a register
//     prepared outside the tile (round 4: the tap decode, select chains and slot counters at the head of the load sections
//     were ~70 instructions of issue against the partner's 258 cycles of MFMAs; unrolled: +13..37 % on every convolution).
//   * PERSISTENT over tiles (round 4): a launch with more tiles than CUs runs one block per CU that walks tiles b, b + G,
//     b + 2G ...  The first patch and the first two weight tiles of the NEXT tile are staged by the current tile's last
//     chunk (they take the places of its "next chunk" patch pieces and of the two weight tiles past the end of K), so only
//     the first tile of a block pays the first-touch latency of its prologue (~6 k cycles of a ~38 k-cycle tile).  The C
//     tile goes through the patch slot the last chunk has just left, in two passes of 128 rows.
//   * logical block row r <-> pixel (r / 16, r % 16) of the tile; the epilogue stores row r there: results are bit-identical
//     to the row-major tiles (tests compare with torch.equal).
#include <type_traits>

#include "common.h"
#include "gemm_args.h"

// tools/lab/gemm8_parts.py builds variants with pieces of the main loop removed (results are garbage): bit 0 no MFMAs,
// 1 no fragment reads, 2 no staging DMAs.  0 in the product build.
#ifndef VN_GEMM8_LAB
#define VN_GEMM8_LAB 0
#endif

namespace {

constexpr int HBM_ROWS = 256, HBN = 128, HNT = 512;
constexpr int PATCH_STRIDE = 41 * 1024;  // 324 pixels x 128 B = 40.5 KiB, rounded up to the 41 wave-DMAs that fill it
constexpr int PROW = 18 * 128;           // bytes per patch row (18 pixels x 64 channels)
constexpr int HB_BYTES = 64 * 128;       // half a B tile: 64 output channels x 64 halfs
constexpr int BBUF = 2 * HB_BYTES;       // a B tile (16 KiB), ring of three
constexpr int BRING = 2 * PATCH_STRIDE;  // the B ring lives behind the two patch slots
constexpr int LOOP_BYTES = BRING + 3 * BBUF;
constexpr int CS_LD = HBN + 8;
constexpr int CS_ROWS = 128;  // C rows per epilogue pass: 128 x 272 B = 34 KiB, inside one patch slot
static_assert(CS_ROWS * CS_LD * 2 <= PATCH_STRIDE, "an epilogue pass must fit the patch slot it reuses");
constexpr int GN_IMG = 5, GN_NG = HBN / 4 + 2;

#define VN_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define VN_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define VN_SYNC(n)                         \
  do {                                     \
    VN_WAIT_VM(n);                         \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
#define VN_BARRIER()                       \
  do {                                     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)

template <int EPI>
__global__ __launch_bounds__(HNT) void halo_conv_kernel(GemmArgs g) {
  const half_t* const e_gate = EPI == 2 ? g.gate_src : nullptr;
  half_t* const e_C2 = EPI == 2 ? g.C2 : nullptr;
  const int e_geglu = EPI == 2 ? g.geglu : 0;
  const int e_act = EPI == 2 ? g.act : 0;
  float* const e_gn_sums = EPI >= 1 ? g.gn_sums : nullptr;
  const half_t* const e_rowadd = EPI >= 1 ? g.rowadd : nullptr;
  const int e_conv = g.conv_mode;
  constexpr int GN_BYTES = EPI == 0 ? 0 : GN_IMG * GN_NG * 4 * 8;  // [image][group][S1.hi S1.lo S2.hi S2.lo] 64-bit words
  __shared__ __attribute__((aligned(16))) char smem[LOOP_BYTES + GN_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2;                     // stagger group
  const int wr4 = wave >> 1, wc2 = wave & 1;    // the wave's 64 x 64 tile in the 4 x 2 grid
  const int frow = lane & 15, fq = lane >> 4, fkey = (frow >> 1) & 7;
  const bool w0 = wave == 0;  // wave 0 stages six patch pieces per chunk, the others five
  const int np_wave = w0 ? 6 : 5;

  const int nblk = g.tiles_m * g.tiles_n;
  const int kz = blockIdx.z;
  const int nk_total = g.K / 64;
  const int kt_begin = kz * g.kt_per_split;
  const int kt_end = min(nk_total, kt_begin + g.kt_per_split);
  const int c_begin = kt_begin / 9, c_end = kt_end / 9;  // this split's channel chunks (the launcher aligns splits to chunks)
  const __amdgpu_buffer_rsrc_t rsA = vn_make_rsrc(g.A, g.a_bytes);
  const __amdgpu_buffer_rsrc_t rsB = vn_make_rsrc(g.B, g.b_bytes);
  const int lrow = tid >> 3;
  const int gchunk = (tid & 7) ^ ((lrow >> 1) & 7);

  // Swizzle key of patch pixel column px (0..17): physical chunk = logical chunk ^ key.  A ds_read_b128 is served in lane
  // groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): eight lanes of one k-chunk and eight of its XOR-1 neighbour,
  // i.e. pixels dx + {0-3, 12-15} with chunk c and dx + {4-11} with chunk c ^ 1.  The row-major tiles' key (px >> 1) & 7
  // is conflict free only for dx = 0: taps with dx = 1, 2 (six of nine) hit two bank groups twice (rocprofv3 in round 4:
  // SQ_LDS_BANK_CONFLICT = 25 % of SQ_LDS_IDX_ACTIVE, 0 on the row-major 256 x 256 tile).  With x_k = key of pixels 2k,
  // 2k + 1 the two window conditions { x0 x1 x2' x3' x4' x5' x6 x7 } and { x1 x2 x3' x4' x5' x6' x7 x8 } (x' = x ^ 1) must
  // both be permutations of 0..7, which forces x6 = x2, x8 = x0; the table is one solution (checked exhaustively over taps,
  // k32 sub-steps and lane groups: tools/lab/patch_swizzle.py).
  auto patch_key = [](const int px) { return (int)((0x270745032ull >> (4 * (px >> 1))) & 7); };

  // ---- tile geometry: tile_m counts 16 x 16 pixel tiles in (image, tile row, tile column) order ----
  struct Tile {
    int tile_m, n0, hb, hty, htx, row0;
  };
  auto tile_of = [&](int vb) {
    // XCD-aware tile mapping (bijective for any tile count): consecutive ids on one XCD sweep N for a fixed M panel
    const int q = nblk >> 3, r = nblk & 7, xcd = vb & 7, idx = vb >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    Tile t;
    t.tile_m = bid / g.tiles_n;
    t.n0 = (bid - t.tile_m * g.tiles_n) * HBN;
    const int tpr = g.Wo >> 4, tpi = (g.Ho >> 4) * tpr;
    t.hb = t.tile_m / tpi;
    const int rem = t.tile_m - t.hb * tpi;
    t.hty = rem / tpr;
    t.htx = rem - t.hty * tpr;
    t.row0 = (t.hb * g.Ho + t.hty * 16) * g.Wo + t.htx * 16;
    return t;
  };
  // ---- patch staging: wave-DMA number d = 8j + wave (d < 41) covers the 64 consecutive 16-byte slots q = 64d + lane of a
  // patch slot; slot q = pixel p = q / 8 (row-major in the 18 x 18 patch), physical chunk q % 8 holding the logical chunk
  // (q % 8) ^ key(px).  Pixels outside the image (the zero padding) and the tail of DMA 40 fetch nothing (out-of-range
  // offset => zeros).  Source offsets of chunk 0; + 128 B per chunk (an out-of-range offset stays out of range: < 2 GiB).
  auto patch_offsets = [&](const Tile& t, uint32_t (&out)[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int q = (j * 8 + wave) * 64 + lane, p = q >> 3, cs = q & 7;
      const int py = (p * 3641) >> 16;  // p / 18 for p < 3000
      const int px = p - py * 18;
      const int iy = t.hty * 16 + py - 1, ix = t.htx * 16 + px - 1;
      const bool ok = p < 324 && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      out[j] = ok ? (uint32_t)(((t.hb * g.Hi + iy) * g.Wi + ix) * g.ldx2 + ((cs ^ patch_key(px)) << 4)) : VN_OOB;
    }
  };
  // weight rows n0 + lrow (first half) / n0 + 64 + lrow (second half) of a B tile, K offset 0
  auto b_offsets = [&](const int n0, uint32_t& h0, uint32_t& h1) {
    const int na = n0 + lrow, nb = n0 + 64 + lrow;
    h0 = na < g.N ? (uint32_t)((long long)na * g.ldb * 2) + gchunk * 16 : VN_OOB;
    h1 = nb < g.N ? (uint32_t)((long long)nb * g.ldb * 2) + gchunk * 16 : VN_OOB;
  };

  // ---- fragment reads: lane (frow = pixel column of the tile row, fq = 8-channel chunk of the k32 sub-step) ----
  int lb[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) lb[dx] = (wr4 * 4) * PROW + (frow + dx) * 128 + ((fq ^ patch_key(frow + dx)) << 4);
  const bool rev = e_conv == 2;
  const int lbr[3] = {lb[2], lb[1], lb[0]};
  const int rb0 = (wc2 * 64 + frow) * 128 + ((fq ^ fkey) << 4);  // (sub-step 1: ^ 64 = chunk + 4)
  char* const bring = smem + BRING + wave * 1024;
  half8 af[4][2], bfh[4][2];
  if constexpr (VN_GEMM8_LAB & 2) {  // (lab build without fragment reads: defined, non-constant operands)
    const half_t v = (half_t)(float)(lane & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i][0] = af[i][1] = bfh[i][0] = bfh[i][1] = half8{v, v, v, v, v, v, v, v};
  }
  auto readAh = [&](const int half, const int ab) {  // tile rows 2 * half, 2 * half + 1 of the wave's four, at tap offset ab
#pragma unroll
    for (int i = 2 * half; i < 2 * half + 2; ++i) {
      if (VN_GEMM8_LAB & 2) continue;
      af[i][0] = as_half8(*reinterpret_cast<const u32x4*>(smem + ab + i * PROW));
      af[i][1] = as_half8(*reinterpret_cast<const u32x4*>(smem + (ab ^ 64) + i * PROW));
    }
  };
  auto readBh = [&](const int slot) {  // the wave's four 16-column blocks of the B tile in ring slot `slot` (compile time)
    const char* b = smem + BRING + slot * BBUF;
    if (VN_GEMM8_LAB & 2) return;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      bfh[cb][0] = as_half8(*reinterpret_cast<const u32x4*>(b + cb * 2048 + rb0));
      bfh[cb][1] = as_half8(*reinterpret_cast<const u32x4*>(b + cb * 2048 + (rb0 ^ 64)));
    }
  };
  // accumulator [i][cb] <-> block row wr4 * 64 + i * 16 + frow, block column wc2 * 64 + cb * 16 + 4 * fq (+ e); operands
  // swapped (D[row = n][col = m]) so a lane owns 4 consecutive n of one m
  auto acc_row = [&](const int i) { return wr4 * 64 + i * 16 + frow; };
  auto acc_col = [&](const int cb) { return wc2 * 64 + cb * 16 + 4 * fq; };
  f32x4 acc[4][4];
  // 16 MFMAs: two row blocks x four column blocks x two k32 sub-steps, at raised priority
#define VN_MMA4(HALF)                                                                                          \
  do {                                                                                                         \
    VN_WAIT_LGKM0();                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                             \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 2 * HALF; i < 2 * HALF + 2; ++i) \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                     \
      if (VN_GEMM8_LAB & 1) {                                                                                  \
        asm volatile("" : "+v"(acc[i][cb]) : "v"(bfh[cb][s]), "v"(af[i][s]));                                  \
      } else {                                                                                                 \
        acc[i][cb] = VN_MFMA_16x16x32(bfh[cb][s], af[i][s], acc[i][cb], 0, 0, 0);                              \
      }                                                                                                        \
    }                                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                                             \
  } while (0)

  // ---- the block's tiles ----
  Tile tl = tile_of(blockIdx.x);
  uint32_t pa_off[6];
  patch_offsets(tl, pa_off);
  uint32_t bb0, bb1;
  b_offsets(tl.n0, bb0, bb1);
  int sbase = 0;  // patch slot of chunk c is (c + sbase) & 1: a tile starts in the slot its predecessor staged for it
  bool first = true;
  for (int vb = blockIdx.x; vb < nblk; vb += gridDim.x) {
    const int row0 = tl.row0, n0 = tl.n0, tile_m = tl.tile_m;
    auto rowmem = [&](const int r) { return row0 + (r >> 4) * g.Wo + (r & 15); };
    // the tile after this one (its first patch and first two B tiles are staged by this tile's last chunk)
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < nblk;
    uint32_t npa[6], nb0 = VN_OOB, nb1 = VN_OOB;
#pragma unroll
    for (int j = 0; j < 6; ++j) npa[j] = VN_OOB;
    if (has_next) {
      const Tile tn = tile_of(nvb);
      patch_offsets(tn, npa);
      b_offsets(tn.n0, nb0, nb1);
    }
    // the bias of this lane's 16 columns, requested before the main loop; out-of-range columns and a null bias read as zeros
    f32x4 bv[4];
    {
      const __amdgpu_buffer_rsrc_t rsBias = vn_make_rsrc(g.bias, g.bias ? (uint32_t)g.N * 4u : 0u);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) bv[cb] = __builtin_bit_cast(f32x4, vn_buf_load16(rsBias, (uint32_t)(n0 + acc_col(cb)) * 4u));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    //   the A fragments of the k-th K-tile of a chunk sit at lb[dx] + dy * PROW of the chunk's slot; the transposed gather
    //            (dgrad) reads patch pixel (y + 2 - dy, x + 2 - dx) = the forward offset of tap 8 - k
    //   bo0/bo1  source offsets of the two halves of the B tile staged next (+128 B per K-tile)
    //   pa_off[j] + cofs   source offset of patch piece j of the NEXT chunk (cofs = 128 B x its chunk index); in the last
    //            chunk npa[j]: piece j of the next tile's first chunk (persistent launches have one K split: chunk 0)
    int slot_off = ((c_begin + sbase) & 1) * PATCH_STRIDE;  // of the chunk being read
    uint32_t bo0 = bb0 + (uint32_t)(kt_begin + 2) * 128u, bo1 = bb1 + (uint32_t)(kt_begin + 2) * 128u;
    // (an out-of-range base stays out of range after the additions: < 2 GiB)
    const uint32_t nbo0 = nb0 + (uint32_t)kt_begin * 128u, nbo1 = nb1 + (uint32_t)kt_begin * 128u;
    if (first) {
      // ---- prologue of the block's first tile: the whole patch of its first chunk, B tiles 0 and 1 ----
      char* p0 = smem + ((c_begin + sbase) & 1) * PATCH_STRIDE + wave * 1024;
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j < np_wave && !(VN_GEMM8_LAB & 4)) dma16(rsA, p0 + j * 8192, pa_off[j] + (uint32_t)c_begin * 128u);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const uint32_t soff = (uint32_t)(kt_begin + kt) * 128u;
        const uint32_t dead = kt_begin + kt < kt_end ? 0u : VN_OOB;
        if (!(VN_GEMM8_LAB & 4)) {
          dma16(rsB, bring + kt * BBUF, (bb0 + soff) | dead);
          dma16(rsB, bring + kt * BBUF + HB_BYTES, (bb1 + soff) | dead);
        }
      }
      VN_WAIT_VM(2);  // the patch and B tile 0 have landed
      VN_BARRIER();
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);
    // K-tile (chunk c, tap): Q0 reads the wave's four column blocks of B(t) and two of its four tile rows at the tap's
    // offset [12 ds_read_b128], stages B(t + 2) [its ring slot was last read in Q0(t - 1)] [+ one DMA of chunk c + 1's patch
    // at taps 1..6: its slot was last read in Q1 of the previous chunk's tap 8, three phases before]; Q1 reads the other two
    // tile rows [4] and waits for B(t + 1) (read in the next Q0): everything but this tile's stagings, which come later in
    // the stream.  In the last chunk the "next chunk" is the NEXT TILE's first chunk and B(T), B(T + 1) are its B tiles 0, 1
    // (out-of-range offsets — nothing fetched — when there is no next tile): the counted waits stay uniform.
    // (Round 4, measured against this loop, all bit-identical: eight reads per phase with column blocks 2, 3 of B(t + 1)
    //  prefetched in Q1 and the wait moved into Q0: equal; every phase's reads issued behind the previous phase's MFMAs,
    //  interleaved with them: +5 %, strictly after them: +9 % slower.)
    auto ktile = [&](auto tap_c, const bool last_chunk, const int pslot, const uint32_t cofs) {
      constexpr int tap = decltype(tap_c)::value;
      constexpr int rslot = tap % 3, sslot = (tap + 2) % 3;  // t % 3 == tap % 3: a chunk is nine K-tiles
      // forward gather: tap (dy, dx) = (tap / 3, tap % 3); transposed gather: the forward offset of tap 8 - tap
      constexpr int dyf = tap / 3, dxf = tap % 3;
      const int ab = (rev ? lbr[dxf] : lb[dxf]) + (rev ? (2 - dyf) * PROW : dyf * PROW) + slot_off;
      // Q0
      readBh(rslot);
      readAh(0, ab);
      if constexpr (tap >= 1 && tap <= 6) {
        if (tap < 6 || w0) {
          if (!(VN_GEMM8_LAB & 4))
            dma16(rsA, smem + pslot + ((tap - 1) * 8) * 1024 + wave * 1024, last_chunk ? npa[tap - 1] : pa_off[tap - 1] + cofs);
        }
      }
      if (!(VN_GEMM8_LAB & 4)) {
        if constexpr (tap >= 7) {
          dma16(rsB, bring + sslot * BBUF, last_chunk ? nbo0 + (tap - 7) * 128u : bo0);
          dma16(rsB, bring + sslot * BBUF + HB_BYTES, last_chunk ? nbo1 + (tap - 7) * 128u : bo1);
        } else {
          dma16(rsB, bring + sslot * BBUF, bo0);
          dma16(rsB, bring + sslot * BBUF + HB_BYTES, bo1);
        }
      }
      bo0 += 128u;
      bo1 += 128u;
      VN_BARRIER();
      VN_MMA4(0);
      VN_BARRIER();
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
      VN_MMA4(1);
      VN_BARRIER();
    };
    for (int c = c_begin; c < c_end; ++c) {
      const bool last_chunk = c + 1 >= c_end;
      const int pslot = ((c + 1 + sbase) & 1) * PATCH_STRIDE;  // where the next chunk's patch goes
      const uint32_t cofs = (uint32_t)(c + 1) * 128u;
      ktile(std::integral_constant<int, 0>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 1>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 2>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 3>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 4>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 5>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 6>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 7>{}, last_chunk, pslot, cofs);
      ktile(std::integral_constant<int, 8>{}, last_chunk, pslot, cofs);
      // the next chunk reads the other patch slot; its successor's pieces come from 64 channels further on
      slot_off = PATCH_STRIDE - slot_off;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger
    // everything staged for the next tile (or the zero-filling stagings past the end) has landed before the epilogue
    // borrows the slot of the last chunk — and before the next tile starts, which therefore has no prologue
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // the C tile's staging area.  Persistent launch: the patch slot the last chunk has just left, in two passes of 128 rows
    // (the other slot and B ring slots 0, 1 hold the next tile's first operands); one tile per block: all 256 rows at once
    // from the bottom of LDS, as the row-major tiles do
    const bool two_pass = gridDim.x < (unsigned)nblk;
    const int pass_rows = two_pass ? CS_ROWS : 2 * CS_ROWS, npass = two_pass ? 2 : 1;
    char* const cs = smem + (two_pass ? ((c_end - 1 + sbase) & 1) * PATCH_STRIDE : 0);

    // ---- split-K: raw f32 partials straight to the workspace (never persistent: one tile per block) ----
    if (g.ksplit > 1) {
      float* ws = g.ws + ((long long)kz * g.M) * g.N;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          const int m = rowmem(acc_row(i));
          const int n = n0 + acc_col(cb);
          if (m < g.M && n < g.N) {
            float* p = ws + (long long)m * g.N + n;
            if (n + 4 <= g.N && (g.N & 3) == 0) {
              *reinterpret_cast<f32x4*>(p) = acc[i][cb];
            } else {
              for (int e = 0; e < 4 && n + e < g.N; ++e) p[e] = acc[i][cb][e];
            }
          }
        }
      return;
    }

    vn_u64* const gacc = reinterpret_cast<vn_u64*>(smem + LOOP_BYTES);
    if (e_gn_sums) {
      for (int i = tid; i < GN_IMG * GN_NG * 4; i += HNT) gacc[i] = 0;
    }
    // ---- epilogue (one pass of 256 rows, or two of 128).  Phase 1: acc -> (alpha, bias, act) -> LDS tile Cs[128][CS_LD] (the waves
    // whose 64 rows lie in the pass).  Rows are 272 B apart (68 dwords = 4 mod 32 banks), so the 16 rows a ds_write_b64 lane
    // group covers would hit every bank pair twice; rows with bit 3 set therefore store the two 8-byte halves of each 16-byte
    // chunk swapped (bank + 2), which phase 2 undoes in registers (there the rows of a thread all share that bit).
    // Phase 2: coalesced row-major stores with the fused operands.  A thread owns one 8-column chunk (column c = 8 * (tid %
    // 16)) of rows tid / 16 + 32 * it, it = 0..3, handled U rows at a time: the C chunks (LDS) and every fused operand
    // (residual, row-add, gate: buffer loads, out-of-range => zeros) of a batch are requested before any of them is used. ----
    half_t* Cb = reinterpret_cast<half_t*>(g.C);
    const half_t* Rb = reinterpret_cast<const half_t*>(g.resid);
    constexpr int CPR = HBN / 8;                 // 16 chunks per row
    constexpr int RPP = HNT / CPR;               // 32 rows per sweep of the block
    constexpr int U = EPI == 2 ? 2 : 4;
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
    const bool swap_halves = ((tid / CPR) >> 3) & 1;  // bit 3 of this thread's rows (the same for all of them, wave-uniform)
    const __amdgpu_buffer_rsrc_t rsR = vn_make_rsrc(Rb, Rb ? 0x7fffffffu : 0u);
    const __amdgpu_buffer_rsrc_t rsRA = vn_make_rsrc(e_rowadd, e_rowadd ? 0x7fffffffu : 0u);
    const __amdgpu_buffer_rsrc_t rsG = vn_make_rsrc(e_gate, e_gate ? 0x7fffffffu : 0u);
    const int r_first = tid / CPR;
    const int nsw = ((frow >> 3) & 1) << 2;  // element offset of the half swap
    // (alpha, bias, act) and the rounding to f16 up front: the f32 accumulators and the bias die here instead of living
    // through the first pass's phase 2 (a wave of the second pass would hold 80 registers through the operand batches)
    half4 oh[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) oh[i][cb][e] = (half_t)apply_act(acc[i][cb][e] * g.alpha + bv[cb][e], e_act);
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      if (!two_pass || (wr4 >> 1) == pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const int ml = acc_row(i) - pass * CS_ROWS;
            const int nl = acc_col(cb) ^ nsw;
            *reinterpret_cast<half4*>(cs + ((size_t)ml * CS_LD + nl) * 2) = oh[i][cb];
          }
      }
      __syncthreads();
      for (int it0 = 0; it0 < pass_rows / RPP; it0 += U) {
        half8 cv[U], rv[U], av[U], gv[U], gv2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rl = r_first + RPP * (it0 + u);
          const int m = rowmem(pass * CS_ROWS + rl);
          const bool ok = m < g.M && full_chunk;
          u32x4 t = *reinterpret_cast<const u32x4*>(cs + ((size_t)rl * CS_LD + c) * 2);
          if (swap_halves) t = u32x4{t[2], t[3], t[0], t[1]};
          cv[u] = as_half8(t);
          if (Rb) rv[u] = as_half8(vn_buf_load16(rsR, ok ? (uint32_t)(((long long)m * g.ldr + n) * 2) : VN_OOB));
          if (EPI >= 1) {
            int grp = 0;
            if (e_rowadd) {
              grp = (int)((float)m * rcp_rpg);  // m / rows_per_group, m < 2^24: the float quotient is off by at most one
              const int rem = m - grp * g.rows_per_group;
              grp += rem >= g.rows_per_group ? 1 : (rem < 0 ? -1 : 0);
            }
            if (e_rowadd) av[u] = as_half8(vn_buf_load16(rsRA, ok ? (uint32_t)(((long long)grp * g.ld_rowadd + n) * 2) : VN_OOB));
          }
          if (EPI == 2 && e_gate) {
            const long long go = (long long)m * g.ld_gate + (e_geglu == 2 ? 2 * n : n);
            gv[u] = as_half8(vn_buf_load16(rsG, ok ? (uint32_t)(go * 2) : VN_OOB));
            gv2[u] = as_half8(vn_buf_load16(rsG, (ok && e_geglu == 2) ? (uint32_t)(go * 2 + 16) : VN_OOB));
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rl = r_first + RPP * (it0 + u);
          const int m = rowmem(pass * CS_ROWS + rl);
          const bool valid = m < g.M && n < g.N;
          if (gn) {
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
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)av[u][e]);
            }
            if (Rb) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rv[u][e]);
            }
            if (e_geglu == 2) {
              // GEGLU backward: v = d(h * gelu(g)) for 8 outputs; the saved pre-activation holds [h0..3 g0..3 h4..7 g4..7]
              half_t* dp = Cb + (long long)m * g.ldc + 2 * n;
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
                *reinterpret_cast<half8*>(dp + 8 * c2) = o;
              }
              continue;
            }
            if (e_gate) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * act_grad((float)gv[u][e], g.gate_act));
            }
            *reinterpret_cast<half8*>(Cb + (long long)m * g.ldc + n) = v;
            if (e_geglu == 1) {
              half4 o2;
#pragma unroll
              for (int e = 0; e < 4; ++e) o2[e] = (half_t)((float)v[e] * vn_gelu_erf((float)v[4 + e]));
              *reinterpret_cast<half4*>(e_C2 + (long long)m * g.ldc2 + (n >> 1)) = o2;
            }
            if (gn) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float x = (float)v[e];
                if (e < gn_split) {
                  s_lo += x;
                  q_lo += x * x;
                } else {
                  s_hi += x;
                  q_hi += x * x;
                }
              }
            }
            if (e_C2 && e_geglu == 0) {
              half8 o2;
#pragma unroll
              for (int e = 0; e < 8; ++e) o2[e] = (half_t)apply_act((float)v[e], g.act2);
              *reinterpret_cast<half8*>(e_C2 + (long long)m * g.ldc2 + n) = o2;
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
      __syncthreads();  // the second pass overwrites the staging area; the next tile restages this slot
    }
    if (gn) {
      gn_flush();
      __syncthreads();
      const int slot = tile_m % g.gn_slots;
      for (int i = tid; i < GN_IMG * GN_NG; i += HNT) {
        const vn_u64* src = gacc + 4 * i;
        if ((src[0] | src[1] | src[2] | src[3]) == 0) continue;
        const int img = gn_img0 + i / GN_NG, grp = gn_g0t + i % GN_NG;
        vn_u64* dst = reinterpret_cast<vn_u64*>(e_gn_sums) + (((long long)img * g.gn_slots + slot) * g.gn_G + grp) * 4;
#pragma unroll
        for (int w = 0; w < 4; ++w) atomicAdd(dst + w, src[w]);
      }
      __syncthreads();  // gacc is cleared at the head of the next tile's epilogue
    }
    // ---- on to the tile this one staged for ----
    sbase = (sbase + (c_end - c_begin)) & 1;
    first = false;
    if (has_next) tl = tile_of(nvb);
#pragma unroll
    for (int j = 0; j < 6; ++j) pa_off[j] = npa[j];
    bb0 = nb0;
    bb1 = nb1;
  }
#undef VN_MMA4
}

inline int epilogue_level_h(const GemmArgs& g) {
  if (g.gate_src || g.C2 || g.geglu || g.act) return 2;
  return (g.gn_sums || g.rowadd) ? 1 : 0;
}

int halo_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            ? p.multiProcessorCount : 256;
  }
  return n;
}

}  // namespace

// f16 output only; the caller (vneti_gemm_f16) has validated the descriptor, set ksplit / kt_per_split and launches the
// split-K reduce itself.  Returns VNETI_EUNSUP for anything but a stride-1 pad-1 3x3 convolution (forward or transposed
// gather) on a 16-pixel grid with chunk-major K; split-K splits are rounded to whole channel chunks.
int vneti_launch_gemm_halo(void* gemm_args, hipStream_t st) {
  GemmArgs& g = *reinterpret_cast<GemmArgs*>(gemm_args);
  if (g.out_f32 || g.ups || g.M >= (1 << 24) || g.conv_mode < 1 || g.conv_mode > 2 || g.stride != 1 || g.pad_t != 1 ||
      g.pad_l != 1 || g.Hi != g.Ho || g.Wi != g.Wo || (g.Ho & 15) || (g.Wo & 15) || (g.Ci & 63) || g.K != 9 * g.Ci ||
      !g.korder || g.batch != 1 || g.M != (g.M / (g.Ho * g.Wo)) * g.Ho * g.Wo)
    return VNETI_EUNSUP;
  if (g.ksplit > 1) {  // splits own whole 64-channel chunks (nine K-tiles each): the patch logic stays per chunk
    const int nchunk = g.Ci / 64, cps = cdiv(nchunk, g.ksplit);
    g.kt_per_split = 9 * cps;
    g.ksplit = cdiv(nchunk, cps);
  }
  g.tiles_m = cdiv(g.M, HBM_ROWS);
  g.tiles_n = cdiv(g.N, HBN);
  const int tiles = g.tiles_m * g.tiles_n;
  // persistent when there is more than one round of tiles: one block per CU walks tiles b, b + G, ... (a multiple of the
  // 8 XCDs, so a block's tiles stay on its XCD's share of the tile order)
  int gx = tiles;
  if (g.ksplit == 1 && tiles > halo_num_cus()) gx = halo_num_cus() & ~7;
  const dim3 grid(gx, 1, g.ksplit), block(HNT);
  const int epi = epilogue_level_h(g);
  if (epi == 2) hipLaunchKernelGGL((halo_conv_kernel<2>), grid, block, 0, st, g);
  else if (epi == 1) hipLaunchKernelGGL((halo_conv_kernel<1>), grid, block, 0, st, g);
  else hipLaunchKernelGGL((halo_conv_kernel<0>), grid, block, 0, st, g);
  return vneti_check_launch("halo_conv_kernel");
}
