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
// Structure: a workgroup of 4, 8 or 16 waves, each wave owning a WM x WN sub-tile built from
// v_mfma_f32_16x16x32_f16; BK = 64; LDS stages of [rows][64 halfs] with the 16-byte chunks
// XOR-swizzled (conflict-free ds_read_b128): two stages, or a 3- / 4-stage ring (NSTG) whose step barrier sits before
// the last MFMA group so the next step's first fragments are fetched under it.  Two ways of filling a stage:
//   DMA  = true : `buffer_load_dwordx4 ... offen lds` (LDS-DMA): no staging VGPRs, no ds_write
//                 pass; the swizzle is applied to the per-lane *source* chunk because the LDS
//                 destination of an LDS-DMA is lane-linear.  Out-of-range offsets return zeros,
//                 which is how conv zero padding and M/N tails are produced.
//   DMA  = false: buffer_load to VGPRs + ds_write_b128 (kept as the A/B reference variant).
// The next tile's loads are in flight under the current tile's MFMAs.  Epilogue staged through
// LDS so HBM stores are full 16-byte rows with bias / time-embedding row-add / residual fused.
// Split-K (grid.z) writes f32 partials to a caller workspace; a second kernel reduces them and
// applies the same epilogue — used for the low-resolution layers whose M is too small to fill
// 256 CUs (M = 256/1024 with K up to 23040).  The kernel is instantiated per epilogue feature group (EPI) and with /
// without the implicit-im2col paths (CONV): code size is a per-launch cost (see the template's comment).
#include "common.h"
#include "gemm_args.h"
#include "../../include/vneti.h"

namespace {

// EPI selects how much of the fused epilogue is compiled in: 0 = bias + residual (most launches), 1 = + time-embedding
// row-add and GroupNorm sums (the resnet convolutions, every VAE conv), 2 = + activation, gate, second output, GEGLU.
// Every runtime-switched feature in the epilogue is paid by every launch (1-2 us x 575 launches per step, DESIGN.md
// section 4) even though the main loops compile to the same instructions.  CONV = false drops the implicit-im2col paths.
template <int BM, int BN, int WM, int WN, bool F32OUT, bool DMA, int NSTG = 2, int EPI = 2, bool CONV = true>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_kernel(
    // the fourteen dwords the prologue needs before its first loads, as explicit parameters: gfx950 delivers those in SGPRs with
    // the wave (kernarg preload, -mllvm -amdgpu-kernarg-preload-count=16 in csrc/build.py); a by-value struct is never
    // preloaded and costs an s_load round trip to cold memory at the head of every launch (tools/lab/kernarg_preload_lab.hip)
    const half_t* pA, const half_t* pB, uint32_t p_a_bytes, uint32_t p_b_bytes, int p_lda, int p_ldb, int pM, int pN, int pK,
    int p_tiles_m, int p_tiles_n, int p_split_conv /* kt_per_split | ksplit << 16 | conv_mode << 24 */, const GemmArgs gfull) {
  GemmArgs g = gfull;  // (scalarised: only the fields a path uses are ever loaded)
  g.A = pA;
  g.B = pB;
  g.a_bytes = p_a_bytes;
  g.b_bytes = p_b_bytes;
  g.lda = p_lda;
  g.ldb = p_ldb;
  g.M = pM;
  g.N = pN;
  g.K = pK;
  g.tiles_m = p_tiles_m;
  g.tiles_n = p_tiles_n;
  g.kt_per_split = p_split_conv & 0xffff;
  g.ksplit = (p_split_conv >> 16) & 0xff;
  g.conv_mode = p_split_conv >> 24;
  static_assert(NSTG == 2 || ((NSTG == 3 || NSTG == 4) && DMA), "the LDS rings are LDS-DMA only");
  const half_t* const e_gate = EPI == 2 ? g.gate_src : nullptr;
  half_t* const e_C2 = EPI == 2 ? g.C2 : nullptr;
  const int e_geglu = EPI == 2 ? g.geglu : 0;
  const int e_act = EPI == 2 ? g.act : 0;
  float* const e_gn_sums = EPI >= 1 ? g.gn_sums : nullptr;
  const half_t* const e_rowadd = EPI >= 1 ? g.rowadd : nullptr;
  const int e_conv = CONV ? g.conv_mode : 0;
  constexpr int NWM = BM / WM, NWN = BN / WN;
  constexpr int NT = NWM * NWN * 64;
  constexpr int RSTEP = NT / 8;  // tile rows covered by one pass of all threads
  constexpr int A_IT = BM / RSTEP, B_IT = BN / RSTEP;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  constexpr int CS_LD = F32OUT ? (BN + 4) : (BN + 8);  // elements
  constexpr int CS_BYTES = BM * CS_LD * (F32OUT ? 4 : 2);
  constexpr int LDS_BYTES = (NSTG * STAGE_BYTES > CS_BYTES) ? NSTG * STAGE_BYTES : CS_BYTES;
  // GroupNorm statistics of the output tile (f16 epilogue only): [GN_IMG images][GN_NG groups][2] floats
  constexpr int GN_IMG = 5, GN_NG = BN / 4 + 2;
  constexpr int GN_BYTES = (F32OUT || EPI == 0) ? 0 : GN_IMG * GN_NG * 4 * 8;  // [image][group][S1.hi S1.lo S2.hi S2.lo] 64-bit words
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES + GN_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int kz = blockIdx.z;

  const half_t* Ab = g.A + (long long)bz * g.strideA;
  const half_t* Bb = g.B + (long long)bz * g.strideB;
  __amdgpu_buffer_rsrc_t rsA = vn_make_rsrc(Ab, g.a_bytes);
  __amdgpu_buffer_rsrc_t rsB = vn_make_rsrc(Bb, g.b_bytes);

  const int lrow = tid >> 3;  // 0..RSTEP-1
  // source chunk fetched by this lane: with LDS-DMA the lane's LDS slot is fixed (lane-linear), so
  // the swizzle moves to the source; with register staging the lane fetches chunk (tid&7) and
  // writes it to the swizzled slot.  (RSTEP is a multiple of 16 => the swizzle key ignores `i`.)
  const int gchunk = DMA ? ((tid & 7) ^ ((lrow >> 1) & 7)) : (tid & 7);

  // ---- per-thread A row bookkeeping (all 32-bit: every tensor is < 2 GiB) -------------------
  const float rcp_hw = CONV ? __builtin_amdgcn_rcpf((float)(g.Ho * g.Wo)) : 0.f, rcp_wo = CONV ? __builtin_amdgcn_rcpf((float)g.Wo) : 0.f;
  int a_off[A_IT];  // plain: row byte offset; conv: byte offset of the (py, px) corner pixel
  int a_py[A_IT], a_px[A_IT], a_bh[A_IT];  // conv: corner coords and b*Hi; a_bh < 0 => invalid row
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + lrow + RSTEP * i;
    bool ok = m < g.M;
    if (e_conv == 0) {
      a_off[i] = ok ? (int)((long long)m * g.lda * 2) + gchunk * 16 : -1;
      a_py[i] = a_px[i] = a_bh[i] = 0;
    } else {
      int rem, ox;  // (b, oy, ox) of output pixel m without integer divisions (common.h vn_divmod)
      const int b = vn_divmod(m, g.Ho * g.Wo, rcp_hw, rem);
      const int oy = vn_divmod(rem, g.Wo, rcp_wo, ox);
      a_bh[i] = ok ? b * g.Hi : -1;
      if (e_conv == 1) {
        a_py[i] = oy * g.stride - g.pad_t;
        a_px[i] = ox * g.stride - g.pad_l;
      } else {
        a_py[i] = oy + g.pad_t;
        a_px[i] = ox + g.pad_l;
      }
      a_off[i] = ((b * g.Hi + a_py[i]) * g.Wi + a_px[i]) * g.ldx2 + gchunk * 16;
    }
  }
  int b_off[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int n = n0 + lrow + RSTEP * i;
    b_off[i] = (n < g.N) ? (int)((long long)n * g.ldb * 2) + gchunk * 16 : -1;
  }

  u32x4 ra[DMA ? 1 : A_IT], rb[DMA ? 1 : B_IT];

  // offsets of tile kt for row slot i (VN_OOB => zeros)
  auto a_offset = [&](int i, int k0, int dy, int dx, int tapoff) -> uint32_t {
    if (e_conv == 0) return a_off[i] < 0 ? VN_OOB : (uint32_t)(a_off[i] + k0 * 2);
    bool ok = a_bh[i] >= 0;
    if (e_conv == 1 && !g.ups) {
      int iy = a_py[i] + dy, ix = a_px[i] + dx;
      ok = ok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      return ok ? (uint32_t)(a_off[i] + tapoff) : VN_OOB;
    }
    int iy, ix;
    if (e_conv == 1) {  // fused nearest-2x upsample
      iy = a_py[i] + dy;
      ix = a_px[i] + dx;
      ok = ok && (unsigned)iy < (unsigned)(2 * g.Hi) && (unsigned)ix < (unsigned)(2 * g.Wi);
      iy >>= 1;
      ix >>= 1;
    } else {  // transposed gather (dgrad)
      int ty = a_py[i] - dy, tx = a_px[i] - dx;
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
    return ok ? (uint32_t)(((a_bh[i] + iy) * g.Wi + ix) * g.ldx2 + tapoff) : VN_OOB;
  };

  // Offsets of the tile about to be fetched.  They are recomputed only when the K index crosses
  // a tap boundary (every Ci/64 tiles, a wave-uniform branch); inside a tap every row just
  // advances by 128 bytes, so the steady-state address cost equals a plain GEMM's.  An
  // out-of-range row stays out of range (VN_OOB + a few KiB never wraps).
  uint32_t a_cur[A_IT], b_cur[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) b_cur[i] = b_off[i] < 0 ? VN_OOB : (uint32_t)b_off[i];

  auto issue = [&](int kt, int stage, bool first) {
    const int k0 = kt * 64;
    bool recompute = first;
    int dy = 0, dx = 0, tapoff = 0;
    if (e_conv != 0) {
      int tap, ci0;
      if (g.korder) {  // chunk-major: consecutive k-steps are the nine taps of one 64-channel chunk
        const int chunk = kt / 9;
        tap = kt - chunk * 9;
        ci0 = chunk << 6;
        recompute = true;
      } else {
        tap = k0 / g.Ci;
        ci0 = k0 - tap * g.Ci;
        recompute = recompute || ci0 == 0;
      }
      if (recompute) {
        dy = tap / 3;
        dx = tap - dy * 3;
        // mode 1 (no upsample): offset relative to the corner pixel; otherwise just channel + chunk
        tapoff = (e_conv == 1 && !g.ups) ? ((dy * g.Wi + dx) * g.ldx2 + ci0 * 2) : (ci0 * 2 + gchunk * 16);
      }
    }
    if (recompute) {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) a_cur[i] = a_offset(i, k0, dy, dx, tapoff);
      if (first) {
#pragma unroll
        for (int i = 0; i < B_IT; ++i) b_cur[i] += (uint32_t)k0 * 2;
      }
    }
    if constexpr (DMA) {
      char* As = smem + stage * STAGE_BYTES;
      char* Bs = As + BM * 128;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) dma16(rsA, As + (wave * 8 + RSTEP * i) * 128, a_cur[i]);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) dma16(rsB, Bs + (wave * 8 + RSTEP * i) * 128, b_cur[i]);
    } else {
#pragma unroll
      for (int i = 0; i < A_IT; ++i) ra[i] = vn_buf_load16(rsA, a_cur[i]);
#pragma unroll
      for (int i = 0; i < B_IT; ++i) rb[i] = vn_buf_load16(rsB, b_cur[i]);
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) a_cur[i] += 128;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_cur[i] += 128;
  };

  auto store_lds = [&](int stage) {
    if constexpr (!DMA) {
      char* As = smem + stage * STAGE_BYTES;
      char* Bs = As + BM * 128;
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        *reinterpret_cast<u32x4*>(As + lds_off(lrow + RSTEP * i, tid & 7)) = ra[i];
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        *reinterpret_cast<u32x4*>(Bs + lds_off(lrow + RSTEP * i, tid & 7)) = rb[i];
    }
  };

  // v_mfma_f32_16x16x32_f16: per flop it moves half the accumulator bytes of the 32x32x16 form and is the more
  // energy-efficient of the two -- at the 1400 W cap that random operands hit, MFMA-only loops run 13 % faster
  // with it (tools/lab/overlap_lab.hip) -- so a wave tile is MI16 x NI16 blocks of 16x16 and a 64-wide k-step is
  // two k32 sub-steps.
  constexpr int MI16 = WM / 16, NI16 = WN / 16;
  f32x4 acc[MI16][NI16];
#pragma unroll
  for (int i = 0; i < MI16; ++i)
#pragma unroll
    for (int j = 0; j < NI16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_total = g.K / 64;
  const int kt_begin = kz * g.kt_per_split;
  const int kt_end = min(nk_total, kt_begin + g.kt_per_split);

  const int frow = lane & 15;  // row of the 16-row fragment this lane feeds
  const int fq = lane >> 4;    // its 8-wide k chunk inside the k32 sub-step / the 4-row group of the result it owns
  half8 af[2][MI16], bf[2][NI16];
  auto load_frags = [&](int buf, int stage, int ks) {
    const char* As = smem + stage * STAGE_BYTES;
    const char* Bs = As + BM * 128;
#pragma unroll
    for (int i = 0; i < MI16; ++i)
      af[buf][i] = as_half8(*reinterpret_cast<const u32x4*>(As + lds_off(wm0 + i * 16 + frow, ks * 4 + fq)));
#pragma unroll
    for (int j = 0; j < NI16; ++j)
      bf[buf][j] = as_half8(*reinterpret_cast<const u32x4*>(Bs + lds_off(wn0 + j * 16 + frow, ks * 4 + fq)));
  };
  auto mma = [&](int buf) {
    if constexpr (NT == 256) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI16; ++i)
#pragma unroll
      for (int j = 0; j < NI16; ++j)
        // operands swapped: D[row = n][col = m]  => each lane owns 4 consecutive n of one m
        acc[i][j] = VN_MFMA_16x16x32(bf[buf][j], af[buf][i], acc[i][j], 0, 0, 0);
    if constexpr (NT == 256) __builtin_amdgcn_s_setprio(0);
  };

  // the bias of this lane's 4 * NI16 columns in one batch of buffer loads (out-of-range columns and a null bias read as
  // zeros), requested BEFORE the K loop: it is cold like everything a launch touches first, and asked for in the epilogue
  // it was a ~1 000-cycle wait on every tile (tools/lab/gemm_stamps.py: accumulators -> LDS 1 700 cycles)
  // (the register-staged 16-wave reference variant has no registers to spare: it asks in the epilogue, as before)
  constexpr bool BIAS_EARLY = DMA || NT < 1024;
  f32x4 bv[NI16];
  auto load_bias = [&]() {
    const __amdgpu_buffer_rsrc_t rsBias = vn_make_rsrc(g.bias, g.bias ? (uint32_t)g.N * 4u : 0u);
#pragma unroll
    for (int j = 0; j < NI16; ++j)
      bv[j] = __builtin_bit_cast(f32x4, vn_buf_load16(rsBias, (uint32_t)(n0 + wn0 + j * 16 + 4 * fq) * 4u));
  };
  if constexpr (BIAS_EARLY) load_bias();

  if constexpr (NSTG == 3) {
    // 3-stage ring with cross-barrier fragment prefetch: the barrier that publishes stage s+1 sits before the
    // LAST MFMA group of step s, so the first fragments of step s+1 are fetched under it and the step boundary
    // has no LDS-latency bubble (tools/lab: +5..24 % on this tile); DMA runs two stages ahead.
    constexpr int PER = A_IT + B_IT;
    issue(kt_begin, 0, true);
    if (kt_begin + 1 < kt_end) {
      issue(kt_begin + 1, 1, false);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    int st = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int st_next = st == 2 ? 0 : st + 1;
      const int st_new = st_next == 2 ? 0 : st_next + 1;
      if (kt + 2 < kt_end) issue(kt + 2, st_new, false);
      load_frags(1, st, 1);
      mma(0);
      if (kt + 2 < kt_end) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < kt_end) load_frags(0, st_next, 0);
      mma(1);
      st = st_next;
    }
    __syncthreads();  // (cheap) the epilogue reuses the ring as its C tile
  } else if constexpr (NSTG == 4) {
    // 4-stage ring for the short-K, latency-bound launches of the small tiles (K = 320 .. 1280 is 5 .. 20 k-steps of
    // ~0.3 us each against ~1.5 us of load latency): three stages in flight instead of two, same cross-barrier
    // fragment prefetch.  DMA of step kt + 3 goes into the stage step kt - 1 has just left.
    constexpr int PER = A_IT + B_IT;
    auto wait_landed = [&](int later) {  // `later` = stages requested after the one that must have landed
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    issue(kt_begin, 0, true);
    if (kt_begin + 1 < kt_end) issue(kt_begin + 1, 1, false);
    if (kt_begin + 2 < kt_end) issue(kt_begin + 2, 2, false);
    wait_landed(kt_end - kt_begin - 1);
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    int st = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int st_next = (st + 1) & 3;
      if (kt + 3 < kt_end) issue(kt + 3, (st + 3) & 3, false);
      load_frags(1, st, 1);
      mma(0);
      wait_landed(kt_end - kt - 2);  // step kt + 1 must be in LDS; kt + 2 and kt + 3 may still be on their way
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < kt_end) load_frags(0, st_next, 0);
      mma(1);
      st = st_next;
    }
    __syncthreads();
  } else {
    issue(kt_begin, 0, true);
    store_lds(0);
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int cur = (kt - kt_begin) & 1;
      if (kt + 1 < kt_end) issue(kt + 1, cur ^ 1, false);
      // fragments double-buffered in registers: the ds_reads of k-step s+1 are in flight under the
      // MFMAs of k-step s; on the 4-wave tiles the MFMA groups run at raised priority so the partner
      // wave's loads/address math do not steal issue slots (+10..23 % in tools/lab).
      load_frags(0, cur, 0);
      load_frags(1, cur, 1);
      mma(0);
      mma(1);
      if (kt + 1 < kt_end) store_lds(cur ^ 1);
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- split-K: raw f32 partials straight to the workspace ---------------------------------
  if (g.ksplit > 1) {
    float* ws = g.ws + ((long long)(kz * g.batch + bz) * g.M) * g.N;
    const __amdgpu_buffer_rsrc_t rsW = vn_make_rsrc(ws, 0x7fffffffu);
#pragma unroll
    for (int i = 0; i < MI16; ++i)
#pragma unroll
      for (int j = 0; j < NI16; ++j) {
        const int m = m0 + wm0 + i * 16 + frow;
        const int n = n0 + wn0 + j * 16 + 4 * fq;
        if (m < g.M && n < g.N) {
          float* p = ws + (long long)m * g.N + n;
          if (n + 4 <= g.N && (g.N & 3) == 0) {
            vn_st16_wt(rsW, (uint32_t)(((long long)m * g.N + n) * 4), acc[i][j]);
          } else {
            for (int e = 0; e < 4 && n + e < g.N; ++e) p[e] = acc[i][j][e];
          }
        }
      }
    return;
  }

  if constexpr (!F32OUT) {
    if (e_gn_sums) {
      vn_u64* gacc = reinterpret_cast<vn_u64*>(smem + LDS_BYTES);
      for (int i = tid; i < GN_IMG * GN_NG * 4; i += NT) gacc[i] = 0;
    }
  }
  // ---- epilogue phase 1: acc -> (alpha, bias, act) -> LDS tile Cs[BM][CS_LD] ---------
  // (the trailing __syncthreads of the K loop guarantees nobody still reads the stages)
  if constexpr (!BIAS_EARLY) load_bias();
#pragma unroll
  for (int i = 0; i < MI16; ++i) {
#pragma unroll
    for (int j = 0; j < NI16; ++j) {
      const int ml = wm0 + i * 16 + frow;
      const int nl = wn0 + j * 16 + 4 * fq;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = acc[i][j][e] * g.alpha + bv[j][e];
        v[e] = apply_act(x, e_act);
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
  __syncthreads();

  // ---- epilogue phase 2: coalesced row-major stores with fused row-add / residual ------
  if constexpr (F32OUT) {
    float* Cb = reinterpret_cast<float*>(g.C) + (long long)bz * g.strideC;
    const __amdgpu_buffer_rsrc_t rsC = vn_make_rsrc(Cb, 0x7fffffffu);
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
        vn_st16_wt(rsC, (uint32_t)(((long long)m * g.ldc + n) * 4), v);
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
    const __amdgpu_buffer_rsrc_t rsC = vn_make_rsrc(Cb, 0x7fffffffu);
    const __amdgpu_buffer_rsrc_t rsC2 = vn_make_rsrc(e_C2, e_C2 ? 0x7fffffffu : 0u);
    const half_t* Rb = reinterpret_cast<const half_t*>(g.resid);
    if (Rb) Rb += (long long)bz * g.strideC;
    constexpr int CPR = BN / 8;
    static_assert(NT % CPR == 0 && CPR <= 32 && (BM * CPR) % NT == 0, "a thread keeps one 8-column chunk over all its rows");
    // GroupNorm statistics of what this tile stores: a thread's chunk touches at most two groups (lo/hi, like
    // csrc/norms.hip); its running sums go to the LDS accumulators whenever the image changes and at the end
    vn_u64* gacc = reinterpret_cast<vn_u64*>(smem + LDS_BYTES);
    const bool gn = e_gn_sums != nullptr;
    const float rcp_gnhw = gn ? __builtin_amdgcn_rcpf((float)g.gn_hw) : 0.f;
    const float rcp_rpg = e_rowadd ? __builtin_amdgcn_rcpf((float)g.rows_per_group) : 0.f;
    const int gn_img0 = gn ? m0 / g.gn_hw : 0, gn_g0t = gn ? n0 / g.gn_cpg : 0;
    const int gn_c = n0 + (tid % CPR) * 8;
    const int gn_glo = gn ? gn_c / g.gn_cpg : 0;
    const int gn_split = gn ? (gn_glo + 1) * g.gn_cpg - gn_c : 8;  // columns [0, split) of the chunk are in group lo
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
    // (a plain rolled loop on purpose.  Its residual / row-add / gate loads are IT serialised round trips — +3 000 cycles on a
    // 19 000-cycle launch of the short-K linears, tools/lab/gemm_stamps.py — but requesting them U rows at a time, as the
    // 8-phase tiles do, needs the loop unrolled: 1 500 cycles faster on those launches, 300 slower on all the others, and
    // 1.5 % slower on the step (a launch's code is cold; its size is a per-launch cost); a one-row-ahead request in the
    // rolled loop gained nothing.  profiles/r04_gemm_epilogue_batch_step_ab.txt)
    for (int idx = tid; idx < BM * CPR; idx += NT) {  // BM * CPR is a multiple of NT: uniform trip count
      int r = idx / CPR, c = (idx - r * CPR) * 8;
      int m = m0 + r, n = n0 + c;
      const bool valid = m < g.M && n < g.N;
      if (gn) {
        int gn_rem;
        const int img = valid ? vn_divmod(m, g.gn_hw, rcp_gnhw, gn_rem) : gn_img;
        if (__any(img != gn_img)) {
          gn_flush();
          gn_img = img;
        }
      }
      if (!valid) continue;
      half8 v = as_half8(*reinterpret_cast<const u32x4*>(smem + ((size_t)r * CS_LD + c) * 2));
      int ra_rem;
      const half_t* radd = e_rowadd ? e_rowadd + (long long)vn_divmod(m, g.rows_per_group, rcp_rpg, ra_rem) * g.ld_rowadd + n : nullptr;
      if (n + 8 <= g.N) {
        if (radd) {
          half8 t = *reinterpret_cast<const half8*>(radd);
          v = vn_add8(v, t);
        }
        if (Rb) {
          half8 rr = *reinterpret_cast<const half8*>(Rb + (long long)m * g.ldr + n);
          v = vn_add8(v, rr);
        }
        if (e_geglu == 2) {
          // GEGLU backward: v = d(h * gelu(g)) for 8 outputs; the saved pre-activation holds [h0..3 g0..3 h4..7 g4..7]
          const half_t* pp = e_gate + (long long)m * g.ld_gate + 2 * n;
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) {
            const half8 pre = *reinterpret_cast<const half8*>(pp + 8 * c2);
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
          half8 pre = *reinterpret_cast<const half8*>(e_gate + (long long)m * g.ld_gate + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * act_grad((float)pre[e], g.gate_act));
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
          vn_st16_wt(rsC2, (uint32_t)(((long long)m * g.ldc2 + n) * 2), o2);
        }
      } else {
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
    if (gn) {
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
}

// split-K second pass: C = epi(alpha * sum_z ws[z]) with the same fused epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(
    // kernarg preload (as gemm_kernel above): what the index arithmetic and the first requests need, in SGPRs with the wave
    float* p_ws, void* pC, const float* p_bias, const void* p_resid, const half_t* p_rowadd, int pM, int pN, int p_batch,
    int p_ksplit, int p_ldc, int p_ldr, const GemmArgs gfull) {
  GemmArgs g = gfull;
  g.ws = p_ws;
  g.C = pC;
  g.bias = p_bias;
  g.resid = p_resid;
  g.rowadd = p_rowadd;
  g.M = pM;
  g.N = pN;
  g.batch = p_batch;
  g.ksplit = p_ksplit;
  g.ldc = p_ldc;
  g.ldr = p_ldr;
  // 32-bit index arithmetic (the launcher checks batch * M * ceil(N / 4) < 2^31): the 64-bit divisions this used to do cost
  // more than the reduction itself on the 110 launches per step, all of them a few microseconds long
  const unsigned n4 = (unsigned)(g.N + 3) / 4;
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const unsigned total = (unsigned)g.batch * (unsigned)g.M * n4;
  if (gid >= total) return;
  const unsigned row = gid / n4;
  const int c = (int)(gid - row * n4) * 4;
  const int bz = g.batch > 1 ? (int)(row / (unsigned)g.M) : 0;
  const int m = (int)(row - (unsigned)bz * (unsigned)g.M);
  const int ne = min(4, g.N - c);
  const long long zstride = (long long)g.batch * g.M * g.N;
  const float* p0 = g.ws + ((long long)bz * g.M + m) * g.N + c;
  if ((g.N & 3) == 0 && ((g.ldc | g.ldr | g.ld_rowadd | g.ld_gate | g.ldc2) & 3) == 0) {
    // vector path (every launch of the step): the epilogue operands are requested first, the partials four at a time --
    // as a scalar loop this kernel was ksplit + 3 serialised load -> wait round trips long
    const long long co = (long long)bz * g.strideC + (long long)m * g.ldc + c;
    const long long ro = (long long)bz * g.strideC + (long long)m * g.ldr + c;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, rf = {0.f, 0.f, 0.f, 0.f};
    half4 rh = {0, 0, 0, 0}, ra = {0, 0, 0, 0}, gt = {0, 0, 0, 0};
    if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + c);
    if (g.resid) {
      if (g.out_f32) rf = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.resid) + ro);
      else rh = *reinterpret_cast<const half4*>(reinterpret_cast<const half_t*>(g.resid) + ro);
    }
    if (g.rowadd) ra = *reinterpret_cast<const half4*>(g.rowadd + (long long)(m / g.rows_per_group) * g.ld_rowadd + c);
    if (g.gate_src) gt = *reinterpret_cast<const half4*>(g.gate_src + (long long)m * g.ld_gate + c);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= g.ksplit; z += 4) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(p0 + (z + 0) * zstride);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(p0 + (z + 1) * zstride);
      const f32x4 t2 = *reinterpret_cast<const f32x4*>(p0 + (z + 2) * zstride);
      const f32x4 t3 = *reinterpret_cast<const f32x4*>(p0 + (z + 3) * zstride);
      v += t0;  // same order as the scalar loop
      v += t1;
      v += t2;
      v += t3;
    }
    for (; z < g.ksplit; ++z) v += *reinterpret_cast<const f32x4*>(p0 + z * zstride);
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[e] = v[e] * g.alpha;
      if (g.bias) x[e] += bv[e];
      x[e] = apply_act(x[e], g.act);
    }
    if (g.out_f32) {
      f32x4 o = {x[0], x[1], x[2], x[3]};
      if (g.resid) o += rf;
      vn_st16_wt(vn_make_rsrc(g.C, 0x7fffffffu), (uint32_t)(co * 4), o);
    } else {
      half4 o, o2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = (float)(half_t)x[e];
        if (g.rowadd) y = (float)(half_t)(y + (float)ra[e]);
        if (g.resid) y = (float)(half_t)(y + (float)rh[e]);
        if (g.gate_src) y = (float)(half_t)(y * act_grad((float)gt[e], g.gate_act));
        o[e] = (half_t)y;
        o2[e] = (half_t)apply_act((float)(half_t)y, g.act2);
      }
      *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(g.C) + co) = o;
      if (g.C2) *reinterpret_cast<half4*>(g.C2 + (long long)m * g.ldc2 + c) = o2;
    }
    return;
  }
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < g.ksplit; ++z) {
    const float* p = p0 + z * zstride;
    for (int e = 0; e < ne; ++e) v[e] += p[e];
  }
  for (int e = 0; e < ne; ++e) {
    float x = v[e] * g.alpha;
    if (g.bias) x += g.bias[c + e];
    x = apply_act(x, g.act);
    if (g.out_f32) {
      if (g.resid) x += reinterpret_cast<const float*>(g.resid)[(long long)bz * g.strideC + (long long)m * g.ldr + c + e];
      reinterpret_cast<float*>(g.C)[(long long)bz * g.strideC + (long long)m * g.ldc + c + e] = x;
    } else {
      x = (float)(half_t)x;
      if (g.rowadd) x = (float)(half_t)(x + (float)g.rowadd[(long long)(m / g.rows_per_group) * g.ld_rowadd + c + e]);
      if (g.resid)
        x = (float)(half_t)(x + (float)reinterpret_cast<const half_t*>(g.resid)[(long long)bz * g.strideC + (long long)m * g.ldr + c + e]);
      if (g.gate_src) x = (float)(half_t)(x * act_grad((float)g.gate_src[(long long)m * g.ld_gate + c + e], g.gate_act));
      reinterpret_cast<half_t*>(g.C)[(long long)bz * g.strideC + (long long)m * g.ldc + c + e] = (half_t)x;
      if (g.C2) g.C2[(long long)m * g.ldc2 + c + e] = (half_t)apply_act((float)(half_t)x, g.act2);
    }
  }
}

// which epilogue instantiation (EPI) a launch needs
inline int epilogue_level(const GemmArgs& g) {
  if (g.gate_src || g.C2 || g.geglu || g.act) return 2;
  return (g.gn_sums || g.rowadd) ? 1 : 0;
}

template <int BM, int BN, int WM, int WN, bool F32OUT, bool DMA, int NSTG>
void launch_variant(const GemmArgs& g, dim3 grid, hipStream_t st) {
  const dim3 block((BM / WM) * (BN / WN) * 64);
#define VN_GO(E, C)                                                                                                        \
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, F32OUT, DMA, NSTG, E, C>), grid, block, 0, st, g.A, g.B, g.a_bytes, g.b_bytes, \
                     (int)g.lda, (int)g.ldb, g.M, g.N, g.K, g.tiles_m, g.tiles_n, g.kt_per_split | (g.ksplit << 16) | (g.conv_mode << 24), g)
  if constexpr (F32OUT) {  // f32 outputs (split-K partials aside: the CLIP residual stream) know bias / act / residual only
    if (g.conv_mode) { if (g.act) VN_GO(2, true); else VN_GO(0, true); }
    else { if (g.act) VN_GO(2, false); else VN_GO(0, false); }
  } else {
    const int epi = epilogue_level(g);
    if (g.conv_mode) { if (epi == 2) VN_GO(2, true); else if (epi == 1) VN_GO(1, true); else VN_GO(0, true); }
    else { if (epi == 2) VN_GO(2, false); else if (epi == 1) VN_GO(1, false); else VN_GO(0, false); }
  }
#undef VN_GO
}

inline void launch_reduce(const GemmArgs& g, hipStream_t st) {
  if (g.ksplit > 1) {
    long long total = (long long)g.batch * g.M * ((g.N + 3) / 4);  // < 2^31: checked with the workspace size in vneti_gemm_f16
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, g.ws, g.C, g.bias, g.resid,
                       g.rowadd, g.M, g.N, g.batch, g.ksplit, (int)g.ldc, (int)g.ldr, g);
  }
}

template <int BM, int BN, int WM, int WN, bool DMA>
int launch_cfg(GemmArgs& g, bool f32out, hipStream_t st) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, BN);
  dim3 grid(g.tiles_m * g.tiles_n, g.batch, g.ksplit);
  if (f32out) launch_variant<BM, BN, WM, WN, true, DMA, 2>(g, grid, st);
  else launch_variant<BM, BN, WM, WN, false, DMA, 2>(g, grid, st);
  launch_reduce(g, st);
  return vneti_check_launch("gemm_kernel");
}

// 3- / 4-stage ring variants (LDS-DMA only)
template <int BM, int BN, int WM, int WN, int NSTG = 3>
int launch_cfg_ring(GemmArgs& g, bool f32out, hipStream_t st) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, BN);
  dim3 grid(g.tiles_m * g.tiles_n, g.batch, g.ksplit);
  if (f32out) launch_variant<BM, BN, WM, WN, true, true, NSTG>(g, grid, st);
  else launch_variant<BM, BN, WM, WN, false, true, NSTG>(g, grid, st);
  launch_reduce(g, st);
  return vneti_check_launch("gemm_kernel");
}

template <int BM, int BN, int WM, int WN, bool DMA>
int launch_cfg_f16(GemmArgs& g, hipStream_t st) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.N, BN);
  dim3 grid(g.tiles_m * g.tiles_n, g.batch, g.ksplit);
  launch_variant<BM, BN, WM, WN, false, DMA, 2>(g, grid, st);
  launch_reduce(g, st);
  return vneti_check_launch("gemm_kernel");
}

struct TileDims {
  int bm, bn;
};
constexpr int kMaxTile = 18;
constexpr TileDims kTiles[kMaxTile + 1] = {{0, 0},     {128, 128}, {128, 64},  {64, 64},  {256, 128}, {256, 256}, {256, 128}, {256, 128},
                                          {256, 128}, {128, 128}, {128, 128}, {128, 64},  {64, 64},   {128, 128}, {128, 64},  {64, 64},
                                          {256, 256}, {256, 128}, {256, 128}};

// tile heuristic: fill >= ~1.5 waves of the 256 CUs when possible, prefer the bigger tile
int select_tile(int M, int N, int batch) {
  long long t256 = (long long)cdiv(M, 256) * cdiv(N, 128) * batch;
  long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128) * batch;
  long long t12864 = (long long)cdiv(M, 128) * cdiv(N, 64) * batch;
  if (N <= 64) return (M >= 2048) ? 2 : 3;
  (void)t256;
  if (t128 >= 384) return 1;
  if (t12864 >= 384) return 2;
  return 3;
}

// split-K factor for a tile choice: only when the grid cannot fill the chip and K is deep
int select_ksplit(int M, int N, int K, int batch, int tile, long long ws_floats) {
  if (ws_floats <= 0) return 1;
  const int nk = K / 64;
  long long tiles = (long long)cdiv(M, kTiles[tile].bm) * cdiv(N, kTiles[tile].bn) * batch;
  if (tiles >= 256 || nk < 16) return 1;
  int s = (int)(512 / tiles);
  if (s > nk / 8) s = nk / 8;
  if (s > 16) s = 16;
  while (s > 1 && (long long)s * batch * M * N > ws_floats) --s;
  return s < 2 ? 1 : s;
}

}  // namespace

extern "C" int vneti_gemm_select_tile(int M, int N, int batch) { return select_tile(M, N, batch > 0 ? batch : 1); }

extern "C" int vneti_gemm_select_split(int M, int N, int K, int batch, int tile_hint, long long workspace_bytes) {
  if (batch <= 0) batch = 1;
  int cfg = tile_hint >= 100 ? tile_hint - 100 : tile_hint;
  if (cfg == 0) cfg = select_tile(M, N, batch);
  if (cfg < 1 || cfg > kMaxTile || K % 64 != 0) return -1;
  int ks = select_ksplit(M, N, K, batch, cfg, workspace_bytes / 4);
  const int nk = K / 64;
  if (ks > nk) ks = nk;
  if (ks < 1) ks = 1;
  return cdiv(nk, cdiv(nk, ks));
}

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
  g.gate_src = (const half_t*)d->gate_src;
  g.ld_gate = d->ld_gate;
  g.gate_act = d->gate_act;
  g.C2 = (half_t*)d->C2;
  g.ldc2 = d->ldc2;
  g.act2 = d->act2;
  g.gn_sums = (float*)d->gn_sums;  // 64-bit fixed-point words (common.h); the kernels reinterpret
  g.gn_hw = d->gn_hw;
  g.gn_cpg = d->gn_cpg;
  g.gn_G = d->gn_groups;
  g.gn_slots = d->gn_slots;
  g.geglu = d->geglu;
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
  g.batch = batch;
  g.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1;
  g.alpha = d->alpha;
  g.act = d->act;
  g.out_f32 = d->out_f32 ? 1 : 0;
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
    VN_REQUIRE(!(d->ups && d->stride != 1), "conv: fused upsample needs stride 1");
    g.Hi = d->Hi;
    g.Wi = d->Wi;
    g.Ci = d->Ci;
    g.Ho = d->Ho;
    g.Wo = d->Wo;
    g.stride = d->stride;
    g.pad_t = d->pad_t;
    g.pad_l = d->pad_l;
    g.ups = d->ups;
    g.ldx2 = (int)(d->ldx * 2);
    // chunk-major K walks (Ci / 64) chunks x 9 taps x 64 channels: a ragged last chunk would read the next tap's weights
    VN_REQUIRE(!d->conv_korder || d->Ci % 64 == 0, "conv: conv_korder=1 needs Ci=%d to be a multiple of 64", d->Ci);
    g.korder = d->conv_korder ? 1 : 0;
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
  if (d->gate_src || d->C2) {
    VN_REQUIRE(!f32 && batch == 1, "gemm: gate/C2 epilogues need f16 output and batch 1");
    VN_REQUIRE(!d->gate_src || (d->gate_act >= 1 && d->gate_act <= 3 && d->ld_gate % 8 == 0), "gemm: bad gate arguments");
    VN_REQUIRE(!d->C2 || (d->act2 >= 0 && d->act2 <= 3 && d->ldc2 % 8 == 0), "gemm: bad C2 arguments");
  }

  if (d->geglu) {
    VN_REQUIRE(d->geglu == 1 || d->geglu == 2, "gemm: geglu must be 0, 1 or 2");
    VN_REQUIRE(!f32 && batch == 1 && d->N % 8 == 0, "gemm: geglu needs f16 output, batch 1 and N %% 8 == 0");
    VN_REQUIRE(d->split_k == 1, "gemm: geglu needs split_k = 1 (it lives in the fused epilogue)");
    VN_REQUIRE(!d->gn_sums, "gemm: geglu and gn_sums are exclusive");
    if (d->geglu == 1) {
      VN_REQUIRE(d->C2 && d->ldc2 % 4 == 0 && !d->gate_src, "gemm: geglu forward needs C2 ([M][N/2]) and no gate");
    } else {
      VN_REQUIRE(d->gate_src && d->ld_gate % 8 == 0 && !d->C2 && !d->resid && !d->rowadd,
                 "gemm: geglu backward needs gate_src ([M][2N] saved pre-activation) and a plain epilogue");
      VN_REQUIRE(d->ldc % 8 == 0, "gemm: geglu backward writes [M][2N]: ldc %% 8");
    }
  }
  if (d->gn_sums) {
    VN_REQUIRE(!f32 && batch == 1, "gemm: gn_sums needs f16 output and batch 1");
    VN_REQUIRE(d->gn_hw >= 64 && d->gn_cpg >= 4 && d->gn_groups > 0 && d->gn_slots > 0 && d->N == d->gn_cpg * d->gn_groups &&
                   d->M % d->gn_hw == 0 && (d->gn_cpg >= 8 || d->gn_cpg == 4),
               "gemm: bad gn_sums geometry hw=%d cpg=%d G=%d", d->gn_hw, d->gn_cpg, d->gn_groups);
    VN_REQUIRE(d->split_k == 1, "gemm: gn_sums needs split_k = 1 (the statistics live in the fused epilogue)");
  }

  int cfg = d->tile_hint;
  bool dma = true;
  if (cfg >= 100) {  // 10x: register-staged reference variant
    dma = false;
    cfg -= 100;
  }
  if (cfg == 0) cfg = select_tile(d->M, d->N, batch);
  VN_REQUIRE(cfg >= 1 && cfg <= kMaxTile, "gemm: unknown tile_hint %d", d->tile_hint);
  // the 8-phase tiles (16 / 17): LDS-DMA only; convolutions whose gather offset is linear in the tap (no fused upsample, no
  // stride-2 transposed gather), either K order (conv_korder 0 tap-major or 1 chunk-major: the offsets are linear in both)
  // the halo-patch form of the 256x128 8-phase tile: stride-1 pad-1 3x3 convolutions (forward or transposed gather) on a
  // 16-pixel grid, chunk-major K (split-K in whole channel chunks); anything else runs as the row-major tile
  if (cfg == 18 &&
      (d->conv_mode < 1 || d->conv_mode > 2 || d->stride != 1 || d->ups || d->pad_t != 1 || d->pad_l != 1 || d->Hi != d->Ho ||
       d->Wi != d->Wo || (d->Ho & 15) || (d->Wo & 15) || (d->Ci & 63) || !d->conv_korder || batch != 1 || f32))
    cfg = 17;
  if (cfg == 18 && d->gn_sums && (d->gn_cpg & 1)) cfg = 17;
  if ((cfg == 16 || cfg == 17) && (!dma || (d->conv_mode && (d->ups || (d->conv_mode == 2 && d->stride == 2))) ||
                    (d->gn_sums && (d->gn_cpg & 1)) ||  // the 8-phase tiles sum the statistics two channels at a time
                    (d->M >= (1 << 24) && (d->conv_mode || d->rowadd || d->gn_sums)))) cfg = cfg == 16 ? 5 : 7;
  if ((cfg == 5 || cfg == 16) && f32) cfg = 4;
  if (cfg == 17 && f32) cfg = 7;  // the 256x256 tiles' f32 epilogue staging would not fit in LDS
  if ((cfg == 6 || cfg == 7) && !dma) cfg = 4;  // the 3-stage ring exists with LDS-DMA only
  if (cfg >= 10 && !dma) cfg = kTiles[cfg].bm == 128 ? (kTiles[cfg].bn == 128 ? 1 : 2) : 3;
  long long ws_floats = d->workspace ? d->workspace_bytes / 4 : 0;
  int ks = d->split_k;
  if (ks == 0) ks = select_ksplit(d->M, d->N, d->K, batch, cfg, ws_floats);
  if (ks < 1) ks = 1;
  const int nk = d->K / 64;
  if (ks > nk) ks = nk;
  if (ks > 1) {
    VN_REQUIRE(d->workspace && (long long)ks * batch * d->M * d->N <= ws_floats,
               "gemm: split_k=%d needs %lld workspace bytes", ks, (long long)ks * batch * d->M * d->N * 4);
    VN_REQUIRE(batch == 1 || (d->strideC != 0), "gemm: batched split-K needs strideC");
    VN_REQUIRE((long long)batch * d->M * ((d->N + 3) / 4) < 0x7fffffffLL, "gemm: split-K output larger than 2^31 chunks");
  }
  if (cfg == 18 && ks > d->Ci / 64) ks = d->Ci / 64;  // the halo form splits in whole 64-channel chunks
  g.ksplit = ks;
  g.kt_per_split = cdiv(nk, ks);
  g.ksplit = cdiv(nk, g.kt_per_split);  // drop empty trailing splits
  VN_REQUIRE(g.ksplit <= 255 && g.kt_per_split <= 0xffff, "gemm: split_k=%d / K=%d out of range", g.ksplit, d->K);
  g.ws = (float*)d->workspace;

#define LAUNCH(BM, BN, WM, WN) \
  return dma ? launch_cfg<BM, BN, WM, WN, true>(g, f32, st) : launch_cfg<BM, BN, WM, WN, false>(g, f32, st)
  switch (cfg) {
    case 1: LAUNCH(128, 128, 64, 64);
    case 2: LAUNCH(128, 64, 64, 32);
    case 3: LAUNCH(64, 64, 32, 32);
    case 4: LAUNCH(256, 128, 64, 64);
    case 6: return launch_cfg_ring<256, 128, 64, 64>(g, f32, st);
    // narrower wave tiles = more waves per SIMD: the loop is latency- rather than bandwidth-limited
    case 7: return launch_cfg_ring<256, 128, 64, 32>(g, f32, st);  // 16 waves x (64x32), 3-stage ring
    case 8: LAUNCH(256, 128, 64, 32);                              // 16 waves x (64x32)
    case 9: LAUNCH(128, 128, 64, 32);                              // 8 waves x (64x32)
    // 3-stage rings of the small tiles: two stages in flight for the fill-bound shapes
    case 10: return launch_cfg_ring<128, 128, 64, 32>(g, f32, st);
    case 11: return launch_cfg_ring<128, 64, 64, 32>(g, f32, st);
    case 12: return launch_cfg_ring<64, 64, 32, 32>(g, f32, st);
    // 4-stage rings: three stages in flight for the short-K launches
    case 13: return launch_cfg_ring<128, 128, 64, 32, 4>(g, f32, st);
    case 14: return launch_cfg_ring<128, 64, 64, 32, 4>(g, f32, st);
    case 15: return launch_cfg_ring<64, 64, 32, 32, 4>(g, f32, st);
    case 16:    // 256x256 as 8 waves in the 8-phase ping-pong structure (gemm8.hip)
    case 17:    // 256x128, same waves, three K-tile buffers
    case 18: {  // 256x128 over a 16 x 16-pixel tile with the input patch resident in LDS (3x3 convolutions)
      const int rc = vneti_launch_gemm8(&g, cfg == 16 ? 256 : 128, cfg == 18 ? 1 : 0, st);
      if (rc != VNETI_OK) return rc;
      launch_reduce(g, st);
      return vneti_check_launch("gemm8_kernel");
    }
    default:
      return dma ? launch_cfg_f16<256, 256, 64, 64, true>(g, st) : launch_cfg_f16<256, 256, 64, 64, false>(g, st);
  }
#undef LAUNCH
}
