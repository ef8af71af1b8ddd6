// Row-stationary persistent linear kernel for gfx950 (tile_hint 19 of vneti_gemm_f16).
//
//   C[M,N] = epilogue( alpha * LN?(A)[M,K] . B[N,K]^T )        K <= 768, K % 64 == 0, f16 in, f32 accumulate, f16 out
//
// The short-K linears of the transformer blocks (to_q / to_k / to_v / to_out of models/xti_attention_processor.py:30-55,
// the GEGLU feed-forward and proj_in / proj_out of diffusers' BasicTransformerBlock, the CLIP MLP) are 5 .. 12 k-tiles
// deep: in the tiled kernels (gemm_conv.hip) a block lives ~19 000 cycles of which ~5 000 issue MFMAs — ring fill, first
// touch, one barrier + 32 blocking LDS-DMA issues per k-step, and an epilogue nobody overlaps (DESIGN.md section 10,
// profiles/r04_gemm128_stamps_before.txt).  This kernel removes the k-loop barrier instead of tuning it:
//
//   * a block owns 64 rows of A for its WHOLE K: the [64][K] tile is fetched once by LDS-DMA (all of it in flight at once,
//     same XOR-swizzled [k-tile][64 rows][128 B] image as the tiled kernels, so fragment reads are conflict-free) and never
//     restaged.  ONE __syncthreads() per block (two with the fused LayerNorm);
//   * each of the 8 waves then sweeps its own column sub-tiles (16 * NF columns) of the block's column range, independently
//     of the other waves: B fragments come straight from global memory / L2 into registers in MFMA operand layout (each
//     re-requested for the next 64-wide k-tile right after its last use), A fragments from the resident tile (4 ds_read_b128 per 4 * NF MFMAs:
//     205 .. 512 LDS bytes per MFMA against 768 for the 128x128 / 8-wave tile), accumulators in registers;
//   * the epilogue of a sub-tile is wave-private (a 32-row LDS staging strip per wave, no barrier): while one wave stores,
//     the other wave of its SIMD issues MFMAs — tile i's stores under tile i+1's loop without a block-wide schedule;
//   * optional fused LayerNorm prologue (ln_gamma != NULL): A is the LayerNorm INPUT; after the tile has landed every row is
//     normalised in place in LDS with the arithmetic of ln_fwd_kernel (csrc/norms.hip: two-pass mean / variance in f32,
//     (x - mean) * rstd * gamma + beta rounded to f16 — what the reference's nn.LayerNorm under fp16 hands the projection),
//     and mean / rstd are published for the backward.  The normalised tensor never exists in HBM and its launch is gone
//     (norm1 -> to_q/k/v, norm2 -> to_q, norm3 -> ff.net.0.proj of every transformer block).
//
// Epilogue features: bias, residual (EPI 0), activation-gradient gate, second activated output, GEGLU forward / backward
// (EPI 2) — the meanings of vneti_gemm_desc, same rounding points as gemm_kernel.  (No `act` on the first output, no
// row-add, no GroupNorm sums: those launches keep their tiled kernels.)
#include <type_traits>

#include "common.h"
#include "gemm_args.h"
#include "../../include/vneti.h"

namespace {

struct LinExtra {
  const float* ln_gamma;
  const float* ln_beta;
  float* ln_mean;
  float* ln_rstd;
  float ln_eps;
  int n_chunks;        // column ranges per row tile (grid = row tiles x n_chunks)
  int cols_per_chunk;  // multiple of 16
};

constexpr int LIN_BM = 64;
constexpr int LIN_KT_MAX = 12;                              // K <= 768
constexpr int LIN_A_BYTES = LIN_KT_MAX * LIN_BM * 128;      // 96 KiB
constexpr int LIN_WAVES = 8;

template <int NF, bool LN, int EPI>
__global__ __launch_bounds__(512) void lin_kernel(const GemmArgs g, const LinExtra x) {
  constexpr int BM = LIN_BM, MI = BM / 16, WN = NF * 16;
  constexpr int ST_LD = WN * 2 + 16;   // bytes per row of a wave's staging strip (32 rows)
  constexpr int ST_BYTES = 32 * ST_LD;
  constexpr int CPR = NF * 2;          // 16-byte chunks per staged row
  __shared__ __attribute__((aligned(16))) char smem[LIN_A_BYTES + LIN_WAVES * ST_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  const int KT = g.K >> 6;

  const int chunk = blockIdx.x % x.n_chunks;
  const int tile_m = blockIdx.x / x.n_chunks;
  const int m0 = tile_m * BM;
  const int c_begin = chunk * x.cols_per_chunk;
  const int c_end = min(g.N, c_begin + x.cols_per_chunk);
  const int nsub = (c_end - c_begin + WN - 1) / WN;
  const int my_nsub = wave < nsub ? (nsub - wave + LIN_WAVES - 1) / LIN_WAVES : 0;
  const int total = my_nsub * KT;  // stages (sub-tile, k-tile) of this wave

  __amdgpu_buffer_rsrc_t rsA = vn_make_rsrc(g.A, g.a_bytes);
  __amdgpu_buffer_rsrc_t rsB = vn_make_rsrc(g.B, g.b_bytes);

  // ---- the whole [64][K] row tile by LDS-DMA: wave w brings rows 8w .. 8w+7 of every k-tile --------------------------------
  {
    const int r = wave * 8 + (lane >> 3);
    const int gchunk = (lane & 7) ^ ((r >> 1) & 7);  // the swizzle lives on the SOURCE side (the LDS side is lane-linear)
    const int m = m0 + r;
    const uint32_t off = m < g.M ? (uint32_t)((long long)m * g.lda * 2) + gchunk * 16 : VN_OOB;
    for (int kt = 0; kt < KT; ++kt) dma16(rsA, smem + kt * (BM * 128) + wave * 1024, off + kt * 128);
  }

  // ---- B fragment stream: ONE register set, each fragment re-requested right after its last use ---------------------------
  // (a fragment's next-stage load is issued as soon as the four MFMAs that read it are out: a full stage of lead time
  //  without a second register set — two sets selected by the stage's parity cost ~100 extra VGPRs in phi copies)
  half8 bq[2][NF];
  uint32_t boff[NF];
  int pf_kt = 0, pf_sub = wave;  // the (sub-tile, k-tile) the running prefetch fetches
  auto b_rows = [&]() __attribute__((always_inline)) {
    const int n0s = c_begin + pf_sub * WN;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int n = n0s + j * 16 + frow;
      boff[j] = n < c_end ? (uint32_t)((long long)n * g.ldb * 2) + fq * 16 : VN_OOB;  // past the range: zeros, no traffic
    }
  };
  auto b_load = [&](int h, int j) __attribute__((always_inline)) {
    return as_half8(__builtin_amdgcn_raw_buffer_load_b128(rsB, boff[j] + 64 * h, pf_kt * 128, 0));
  };
  auto b_advance = [&]() __attribute__((always_inline)) {  // the prefetch stream moves on to the next stage
    if (++pf_kt == KT) {
      pf_kt = 0;
      pf_sub += LIN_WAVES;
      b_rows();
    }
  };
  if (total > 0) {
    b_rows();
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < NF; ++j) bq[h][j] = b_load(h, j);
    b_advance();
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- fused LayerNorm: normalise the resident tile in place (8 threads per row, one 16-byte chunk column each) ----------
  if constexpr (LN) {
    const int r = tid >> 3, sub = tid & 7;
    const int jc = sub ^ ((r >> 1) & 7);  // the k-chunk this LDS position holds
    char* rowp = smem + r * 128 + sub * 16;
    half8 v[LIN_KT_MAX];
    float s = 0.f;
#pragma unroll
    for (int kt = 0; kt < LIN_KT_MAX; ++kt) {
      if (kt < KT) {
        v[kt] = as_half8(*reinterpret_cast<const u32x4*>(rowp + kt * (BM * 128)));
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += (float)v[kt][e];
        s += t;
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    const float inv_c = __builtin_amdgcn_rcpf((float)g.K);
    const float mu = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int kt = 0; kt < LIN_KT_MAX; ++kt) {
      if (kt < KT) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)v[kt][e] - mu;
          t += d * d;
        }
        q += t;
      }
    }
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    q += __shfl_xor(q, 4);
    const float rs = rsqrtf(q * inv_c + x.ln_eps);
    if (sub == 0 && chunk == 0 && m0 + r < g.M) {
      if (x.ln_mean) x.ln_mean[m0 + r] = mu;
      if (x.ln_rstd) x.ln_rstd[m0 + r] = rs;
    }
#pragma unroll
    for (int kt = 0; kt < LIN_KT_MAX; ++kt) {
      if (kt < KT) {
        const float* gp = x.ln_gamma + kt * 64 + jc * 8;
        const float* bp = x.ln_beta + kt * 64 + jc * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
        half8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (half_t)(((float)v[kt][e] - mu) * rs * g0[e] + b0[e]);
          o[4 + e] = (half_t)(((float)v[kt][4 + e] - mu) * rs * g1[e] + b1[e]);
        }
        *reinterpret_cast<u32x4*>(rowp + kt * (BM * 128)) = as_u32x4(o);
      }
    }
    __syncthreads();
  }
  if (total == 0) return;

  // ---- per-lane constants of the sweep -----------------------------------------------------------------------------------
  // fragment rows i * 16 + frow share one swizzle key ((row >> 1) & 7 ignores multiples of 16): one offset per k32 half,
  // the four row blocks are immediates
  const int aoff0 = lds_off(frow, fq), aoff1 = lds_off(frow, 4 + fq);
  char* const stg = smem + LIN_A_BYTES + wave * ST_BYTES;
  const __amdgpu_buffer_rsrc_t rsBias = vn_make_rsrc(g.bias, g.bias ? (uint32_t)g.N * 4u : 0u);
  const half_t* const e_gate = EPI == 2 ? g.gate_src : nullptr;
  half_t* const e_C2 = EPI == 2 ? g.C2 : nullptr;
  const int e_geglu = EPI == 2 ? g.geglu : 0;
  half_t* const Cb = reinterpret_cast<half_t*>(g.C);
  const half_t* const Rb = reinterpret_cast<const half_t*>(g.resid);

  f32x4 acc[MI][NF];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int sub = wave;  // the sub-tile being computed

  // the epilogue of one finished sub-tile: accumulators -> (alpha, bias, act) -> f16 -> the wave's staging strip -> 16-byte
  // row chunks with the fused operands -> global.  LDS operations of one wave execute in order: no barrier.
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int n_sub0 = c_begin + sub * WN;
    f32x4 bv[NF];  // (an L2 round trip per sub-tile; the SIMD's other wave is in its loop meanwhile)
#pragma unroll
    for (int j = 0; j < NF; ++j)
      bv[j] = __builtin_bit_cast(f32x4, vn_buf_load16(rsBias, (uint32_t)(n_sub0 + j * 16 + 4 * fq) * 4u));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int i = half * 2 + i2;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          half4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(acc[i][j][e] * g.alpha + bv[j][e]);
          *reinterpret_cast<u32x2*>(stg + (i2 * 16 + frow) * ST_LD + (j * 16 + 4 * fq) * 2) = as_u32x2(o);
        }
      }
#pragma nounroll  // (one copy of the row pass per half: a launch's code is cold, its size is a per-launch cost)
      for (int t = 0; t < (32 * CPR) / 64; ++t) {
        const int idx = lane + 64 * t;
        const int r = idx / CPR, c = idx - r * CPR;
        const int m = m0 + half * 32 + r, n = n_sub0 + c * 8;
        half8 v = as_half8(*reinterpret_cast<const u32x4*>(stg + r * ST_LD + c * 16));
        if (m >= g.M || n >= c_end) continue;
        if (Rb) v = vn_add8(v, *reinterpret_cast<const half8*>(Rb + (long long)m * g.ldr + n));
        if constexpr (EPI == 2) {
          if (e_geglu == 2) {
            // GEGLU backward: v = d(h * gelu(g)) for 8 outputs; the saved pre-activation holds [h0..3 g0..3 h4..7 g4..7]
            const half_t* pp = e_gate + (long long)m * g.ld_gate + 2 * n;
            half_t* dp = Cb + (long long)m * g.ldc + 2 * n;
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
              *reinterpret_cast<half8*>(dp + 8 * c2) = o;
            }
            continue;
          }
          if (e_gate) {
            const half8 pre = *reinterpret_cast<const half8*>(e_gate + (long long)m * g.ld_gate + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * act_grad((float)pre[e], g.gate_act));
          }
        }
        *reinterpret_cast<half8*>(Cb + (long long)m * g.ldc + n) = v;
        if constexpr (EPI == 2) {
          if (e_geglu == 1) {
            half4 o2;
#pragma unroll
            for (int e = 0; e < 4; ++e) o2[e] = (half_t)((float)v[e] * vn_gelu_erf((float)v[4 + e]));
            *reinterpret_cast<half4*>(e_C2 + (long long)m * g.ldc2 + (n >> 1)) = o2;
          } else if (e_C2) {
            half8 o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = (half_t)apply_act((float)v[e], g.act2);
            *reinterpret_cast<half8*>(e_C2 + (long long)m * g.ldc2 + n) = o2;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // one stage = 64 of K for the wave's 64 x (16 * NF) sub-tile.  Per column block: eight MFMAs (both k32 halves), then the
  // request for its two next-stage fragments TOGETHER — they are the two halves of the same 128-byte lines of 16 weight
  // rows, so the second load finds the lines the first one has just asked for (requested half a stage apart, the other half
  // came back from L2 a second time: twice the L2 -> L1 traffic of a kernel whose cost IS that traffic).  Unconditional:
  // past the wave's last stage the offsets are out of range.  A fragments are read one stage ahead (two register buffers).
  // The order is pinned with sched_barrier — left alone, hipcc sinks all loads below the last MFMA of the stage and the
  // stage of lead time shrinks to a third.  k order inside a row is the tiled kernels': results are bit-identical to theirs.
#define VN_SB() __builtin_amdgcn_sched_barrier(0)
  half8 af[2][2][MI];  // [buffer][half][row block]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < MI; ++i) af[0][h][i] = as_half8(*reinterpret_cast<const u32x4*>(smem + (h ? aoff1 : aoff0) + i * 2048));
  auto stage = [&](auto cur_c, int kt) __attribute__((always_inline)) {
    constexpr int cur = decltype(cur_c)::value;
    const char* An = smem + (kt + 1 == KT ? 0 : kt + 1) * (BM * 128);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[cur ^ 1][h][i] = as_half8(*reinterpret_cast<const u32x4*>(An + (h ? aoff1 : aoff0) + i * 2048));
    VN_SB();
#pragma unroll
    for (int j = 0; j < NF; ++j) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          // operands swapped: D[row = n][col = m] => a lane owns 4 consecutive n of one m
          acc[i][j] = VN_MFMA_16x16x32(bq[h][j], af[cur][h][i], acc[i][j], 0, 0, 0);
      bq[0][j] = b_load(0, j);
      bq[1][j] = b_load(1, j);
      VN_SB();
    }
    b_advance();
  };
  // (the A fragment buffers alternate by stage: K / 64 may be odd, so the k loop is unrolled by two with a tail)
  for (int s = 0; s < my_nsub; ++s) {
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
      stage(std::integral_constant<int, 0>{}, kt);
      stage(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < KT) {  // odd K / 64: the tail stage leaves the next sub-tile's first fragments in buffer 1
      stage(std::integral_constant<int, 0>{}, kt);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < MI; ++i) af[0][h][i] = af[1][h][i];
    }
    epilogue();
    sub += LIN_WAVES;
  }
#undef VN_SB
}

// cost model of one (NF, n_chunks) choice, in k-tile units of one wave's work (deterministic: every rank picks alike)
struct LinPlan {
  int nf, chunks, cols;
  double cost;
};

LinPlan lin_plan(int M, int N, int K) {
  const int m_tiles = cdiv(M, LIN_BM);
  const int kt = K / 64;
  LinPlan best{0, 0, 0, 1e30};
  for (int nf = 5; nf >= 2; --nf)
    for (int chunks = 1; chunks <= 16; ++chunks) {
      int cols = cdiv(cdiv(N, chunks), 16) * 16;
      if (chunks > 1 && cols < 16 * nf * LIN_WAVES) continue;  // a chunk narrower than one round of sub-tiles
      if ((long long)cols * (chunks - 1) >= N) continue;       // empty last chunk
      const int nsub = cdiv(cols, 16 * nf);
      const int rounds = cdiv(nsub, LIN_WAVES);
      const long long blocks = (long long)m_tiles * chunks;
      const double waves_of_blocks = (double)cdivl(blocks, 256);
      // per sub-tile and k-tile: 8 * nf MFMAs (16 cycles each, two waves share a SIMD) against 8 fragment reads; the
      // row tile's fill + (amortised) epilogue as a fixed part
      const double per_sub = kt * (2.0 * nf + 3.0) + 6.0 + 2.0 * nf;
      const double block = rounds * per_sub + 2.0 * kt + 8.0;
      const double cost = waves_of_blocks * block;
      if (cost < best.cost) best = LinPlan{nf, chunks, cols, cost};
    }
  return best;
}

template <int NF>
void lin_launch_nf(const GemmArgs& g, const LinExtra& x, int epi, dim3 grid, hipStream_t st) {
  const bool ln = x.ln_gamma != nullptr;
#define VN_LIN(L, E) hipLaunchKernelGGL((lin_kernel<NF, L, E>), grid, dim3(512), 0, st, g, x)
  if (ln) {
    if (epi) VN_LIN(true, 2); else VN_LIN(true, 0);
  } else {
    if (epi) VN_LIN(false, 2); else VN_LIN(false, 0);
  }
#undef VN_LIN
}

}  // namespace

// can tile_hint 19 run this problem?  (plain f16 GEMM, one batch, short K; no row-add, no GroupNorm sums, no split-K)
int vneti_linear_eligible(const vneti_gemm_desc* d) {
  const int batch = d->batch > 1 ? d->batch : 1;
  return d->conv_mode == 0 && batch == 1 && !d->out_f32 && d->K % 64 == 0 && d->K <= 64 * LIN_KT_MAX && d->K >= 64 &&
         d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && d->ldc % 8 == 0 && !d->rowadd && !d->gn_sums && !d->act &&
         (!d->resid || d->ldr % 8 == 0) && d->M > 0 && d->N >= 16;
}

int vneti_launch_linear(void* gemm_args, const vneti_gemm_desc* d, hipStream_t st) {
  GemmArgs& g = *reinterpret_cast<GemmArgs*>(gemm_args);
  LinExtra x{};
  x.ln_gamma = d->ln_gamma;
  x.ln_beta = d->ln_beta;
  x.ln_mean = d->ln_mean;
  x.ln_rstd = d->ln_rstd;
  x.ln_eps = d->ln_eps;
  VN_REQUIRE(!x.ln_gamma || x.ln_beta, "linear: ln_gamma without ln_beta");
  const LinPlan p = lin_plan(g.M, g.N, g.K);
  VN_REQUIRE(p.nf >= 2, "linear: no plan for M=%d N=%d K=%d", g.M, g.N, g.K);
  x.n_chunks = p.chunks;
  x.cols_per_chunk = p.cols;
  g.ksplit = 1;
  const int epi = (g.gate_src || g.C2 || g.geglu) ? 1 : 0;
  dim3 grid((unsigned)(cdiv(g.M, LIN_BM) * p.chunks));
  switch (p.nf) {
    case 5: lin_launch_nf<5>(g, x, epi, grid, st); break;
    case 4: lin_launch_nf<4>(g, x, epi, grid, st); break;
    case 3: lin_launch_nf<3>(g, x, epi, grid, st); break;
    default: lin_launch_nf<2>(g, x, epi, grid, st); break;
  }
  return vneti_check_launch("lin_kernel");
}
