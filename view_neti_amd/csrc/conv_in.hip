// 3x3 / stride 1 / pad 1 convolution of an image with <= 3 channels straight from the (arbitrarily strided, f32 or f16)
// pixels: AutoencoderKL.encoder.conv_in (diffusers vae.py Encoder.conv_in, driven from training/coach.py:165-169).
//
// The layer is pure output bandwidth: 27 MACs per output element, 256 B written per pixel for 24 B read.  The generic path
// (vneti_im2col3x3_small + GEMM with K padded 27 -> 64) writes and re-reads a [pixels][64] f16 matrix four times the size
// of the image before the first MFMA; here a wave builds the [16 pixels][32] operand in registers (k = tap * C + c, zeros
// from 9 * C up), multiplies it against the register-resident weights with v_mfma_f32_16x16x32_f16 (one k-step) and
// stores 16-byte chunks.  The packed weight rows are permuted (packing.conv_in_direct) so that the four accumulator
// values of MFMA row block j in quarter-wave fq belong to output channels (j/2)*32 + fq*8 + (j%2)*4 + e: a lane then
// owns 8 consecutive channels per 32-channel chunk and the four quarter-wave lanes of a pixel write 64 contiguous bytes
// per store.  GroupNorm (sum, sum of squares) of the stored f16 values are accumulated like the GEMM epilogue does
// (vneti_gemm_desc.gn_sums) for the first ResnetBlock2D.norm1.
#include "common.h"
#include "../../include/vneti.h"

namespace {

struct ConvInArgs {
  const void* x;
  long long sb, sc, sy, sx;
  const half_t* w;  // [Co][32] f16, rows permuted per 128-channel block, k = tap * C + c
  const float* bias;
  half_t* out;
  long long ldo;
  int Bn, C, H, W, Co;
  void* gn_sums;  // 64-bit fixed-point slot sums, see common.h
  int gn_cpg, gn_G, gn_slots;
};

constexpr int PIX_PER_BLOCK = 256;  // 4 waves x 4 groups of 16 pixels

template <bool F32IN>
__global__ __launch_bounds__(256) void conv_in_kernel(ConvInArgs a) {
  __shared__ vn_u64 gacc[32 * 4];  // fixed-point (sum, sumsq) of the 32 channel quads of this 128-channel block (common.h)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fq = lane >> 4;
  const int nb = blockIdx.y;  // 128-channel block
  const long long HW = (long long)a.H * a.W;
  const bool gn = a.gn_sums != nullptr;
  if (tid < 128) gacc[tid] = 0;

  // weights of this channel block: 8 row blocks of 16 x 32 halfs, one 16-byte chunk per lane each
  half8 wf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    wf[j] = *reinterpret_cast<const half8*>(a.w + ((long long)(nb * 128 + j * 16 + frow)) * 32 + fq * 8);
  // bias of the 32 channels this lane ends up with
  f32x4 bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int co = nb * 128 + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
    bv[j] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gs[8], gq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) gs[j] = gq[j] = 0.f;
  __syncthreads();

  const long long m_blk = (long long)blockIdx.x * PIX_PER_BLOCK;
  const long long M = (long long)a.Bn * HW;
  // all four pixel groups' gathers first (32 independent loads in flight per lane), then the MFMAs and stores
  half8 pfs[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const long long m = m_blk + (wave * 4 + g) * 16 + frow;
    const bool mok = m < M;
    const int b = (int)(m / HW);
    const int rem = (int)(m - (long long)b * HW);
    const int y = rem / a.W, xx = rem - y * a.W;
    // this lane's 8 consecutive k of pixel m: k = tap * C + c
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      // branch-free gather: the address is clamped into the image and the value masked afterwards, so the 32 loads of a
      // lane are all in flight together (as `if (in range) load` the compiler waited for every one of them in turn)
      const int k = fq * 8 + t;
      const int tap = k / a.C, c = k - tap * a.C;
      const int iy = y + tap / 3 - 1, ix = xx + tap % 3 - 1;
      const bool in = mok && tap < 9 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int bs = mok ? b : 0, cs = tap < 9 ? c : 0;
      const int ys = min(max(iy, 0), a.H - 1), xs = min(max(ix, 0), a.W - 1);
      const long long off = (long long)bs * a.sb + (long long)cs * a.sc + (long long)ys * a.sy + (long long)xs * a.sx;
      float v;
      if constexpr (F32IN) v = reinterpret_cast<const float*>(a.x)[off];
      else v = (float)reinterpret_cast<const half_t*>(a.x)[off];
      pfs[g][t] = (half_t)(in ? v : 0.f);
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const long long m = m_blk + (wave * 4 + g) * 16 + frow;
    const bool mok = m < M;
    const half8 pf = pfs[g];
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      half8 o;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = c4 * 2 + jj;
        // D[row = channel][col = pixel]: the lane owns rows 4 * fq .. + 3 of row block j for pixel frow
        f32x4 acc = VN_MFMA_16x16x32(wf[j], pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half_t hv = (half_t)(acc[e] + bv[j][e]);
          o[jj * 4 + e] = hv;
          if (gn && mok) {
            const float xf = (float)hv;
            gs[j] += xf;
            gq[j] += xf * xf;
          }
        }
      }
      if (mok)  // write-through (common.h vn_st16_wt): 268 MB of output at 512 x 512 leave L2 while the kernel runs
        vn_st16_wt(vn_make_rsrc(a.out, 0x7fffffffu), (uint32_t)((m * a.ldo + nb * 128 + c4 * 32 + fq * 8) * 2), o);
    }
  }

  if (gn) {  // the block's pixels lie in one image (host: H * W % 256 == 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        gs[j] += __shfl_xor(gs[j], off);
        gq[j] += __shfl_xor(gq[j], off);
      }
      if (frow == 0) {
        const int quad = (j >> 1) * 8 + fq * 2 + (j & 1);  // channel quad inside the 128-channel block
        vn_fx_add2(&gacc[4 * quad], gs[j], gq[j]);  // integer atomics: order-independent totals
      }
    }
    __syncthreads();
    const int qpg = a.gn_cpg >> 2;  // quads per group
    const int ng = 32 / qpg;        // groups in this channel block
    if (tid < ng) {
      vn_u64 t[4] = {0, 0, 0, 0};
      for (int i = 0; i < qpg; ++i)
#pragma unroll
        for (int w = 0; w < 4; ++w) t[w] += gacc[4 * (tid * qpg + i) + w];
      const int img = (int)(m_blk / HW), slot = blockIdx.x % a.gn_slots, grp = nb * ng + tid;
      vn_u64* dst = reinterpret_cast<vn_u64*>(a.gn_sums) + (((long long)img * a.gn_slots + slot) * a.gn_G + grp) * 4;
#pragma unroll
      for (int w = 0; w < 4; ++w) atomicAdd(dst + w, t[w]);
    }
  }
}

}  // namespace

extern "C" int vneti_conv3x3_in(const void* x, int x_is_f32, long long sb, long long sc, long long sy, long long sx,
                                const void* w_packed, const float* bias, void* out, long long ldo, int Bn, int C, int H,
                                int W, int Co, void* gn_sums, int gn_groups, int gn_slots, void* stream) {
  VN_REQUIRE(x && w_packed && out, "conv3x3_in: null pointer");
  VN_REQUIRE(C >= 1 && C <= 3, "conv3x3_in: C=%d (9*C must fit one 32-wide k-step)", C);
  VN_REQUIRE(Co > 0 && Co % 128 == 0, "conv3x3_in: Co=%d must be a multiple of 128", Co);
  VN_REQUIRE(Bn > 0 && H > 0 && W > 0 && ldo % 8 == 0, "conv3x3_in: bad geometry");
  ConvInArgs a;
  a.x = x;
  a.sb = sb;
  a.sc = sc;
  a.sy = sy;
  a.sx = sx;
  a.w = (const half_t*)w_packed;
  a.bias = bias;
  a.out = (half_t*)out;
  a.ldo = ldo;
  a.Bn = Bn;
  a.C = C;
  a.H = H;
  a.W = W;
  a.Co = Co;
  a.gn_sums = gn_sums;
  a.gn_G = gn_groups;
  a.gn_slots = gn_slots;
  a.gn_cpg = gn_groups > 0 ? Co / gn_groups : 0;
  if (gn_sums) {
    VN_REQUIRE(gn_groups > 0 && gn_slots > 0 && Co % gn_groups == 0 && a.gn_cpg % 4 == 0 && a.gn_cpg <= 128 &&
                   128 % a.gn_cpg == 0 && ((long long)H * W) % PIX_PER_BLOCK == 0,
               "conv3x3_in: gn_sums needs cpg in {4, 8, .., 128} and H*W %% 256 == 0");
  }
  const long long M = (long long)Bn * H * W;
  dim3 grid((unsigned)cdivl(M, PIX_PER_BLOCK), Co / 128);
  if (x_is_f32)
    hipLaunchKernelGGL((conv_in_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv_in_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  return vneti_check_launch("conv_in_kernel");
}
