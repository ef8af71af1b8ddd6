// GroupNorm(+SiLU) on channels-last f16 and LayerNorm, forward and input-gradient.
// HBM-bound kernels: 16-byte loads/stores, f32 statistics, deterministic two-level reductions.
//
// Reference call sites: diffusers ResnetBlock2D.norm1/norm2 + SiLU, Transformer2DModel.norm,
// conv_norm_out, BasicTransformerBlock.norm1..3 (all inside `self.unet(...)`,
// training/coach.py:197-198) and CLIPEncoderLayer.layer_norm1/2 + final_layer_norm
// (models/neti_clip_text_encoder.py:111-118,183-185).
#include "common.h"
#include "../../include/vneti.h"

namespace {

// ------------------------------------------------------------------------------------------
// GroupNorm.  x: [B][HW][ldx] with C channels, G groups, cpg = C/G >= 4.
// Work decomposition: block (slab, b) owns `rps` consecutive pixels of sample b.  Thread
// (tx, ty) owns 8-channel chunk `tx` (looping in steps of TX when C/8 > 256) and walks rows
// ty, ty+TY, ...  A chunk touches at most two groups ("lo"/"hi").
// Pass 1 writes per-slab partial sums; pass 2 (one block per sample) finalises mean/rstd in
// f64; pass 3 applies.  The backward has the same three-pass shape.
// ------------------------------------------------------------------------------------------
#ifndef GN_APPLY_U
#define GN_APPLY_U 4
#endif
struct GNGeom {
  int Bn, HW, C, G, cpg, CC, TX, TY, rps, nslab;
  double inv_n;  // 1 / (HW * cpg): the statistics are finished with multiplications, not f64 divisions
};

// mean / rstd from (sum, sum of squares): the cancellation E[x^2] - E[x]^2 in f64, the rest in f32 (v_rsq_f32 + one Newton
// step) — an f64 division and square root per thread cost the apply kernels more than a thousand cycles before their first load
__device__ __forceinline__ void vn_mean_rstd(double s0, double s1, double inv_n, float eps, float& m, float& r) {
  const double mu = s0 * inv_n;
  const double var = s1 * inv_n - mu * mu;
  const float v = fmaxf((float)var, 0.f) + eps;
  float y = __builtin_amdgcn_rsqf(v);
  y = y * (1.5f - 0.5f * v * y * y);
  m = (float)mu;
  r = y;
}

__device__ __forceinline__ void gn_thread_coords(const GNGeom& g, int& tx, int& ty, bool& active) {
  int tid = threadIdx.x;
  tx = tid % g.TX;
  ty = tid / g.TX;
  active = ty < g.TY;
}

// SLOTS = false: this slab's sums go to part[b][slab][G][2] (reduced by gn_finalize_kernel).
// SLOTS = true : they are ADDED to part[b][slab % slots][G][2] (caller-zeroed), the layout the GEMM epilogues use for
//                vneti_gemm_desc.gn_sums, so the apply kernel finishes the statistics itself and the finalize launch goes.
template <bool BWD, bool SILU, bool SLOTS = false>
__global__ __launch_bounds__(256) void gn_stats_kernel(GNGeom g, const half_t* __restrict__ x, long long ldx,
                                                       const half_t* __restrict__ dy, long long lddy,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       float* __restrict__ part, int slots = 0) {
  __shared__ vn_u64 gacc[64 * 4];  // [G <= 64][S1.hi S1.lo S2.hi S2.lo]: fixed-point sums (common.h), order-independent
  const int slab = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < 4 * g.G; i += 256) gacc[i] = 0;
  __syncthreads();
  int tx, ty;
  bool active;
  gn_thread_coords(g, tx, ty, active);
  const int r0 = slab * g.rps;
  const int r1 = min(r0 + g.rps, g.HW);
  if (active) {
    for (int cb = 0; cb < g.CC; cb += g.TX) {
      const int cx = cb + tx;
      if (cx >= g.CC) break;
      const int ch0 = cx * 8;
      const int g0 = ch0 / g.cpg;
      const int split = (g0 + 1) * g.cpg - ch0;  // elements [0,split) belong to g0, rest to g0+1
      float ga[8], be[8];
      float mlo = 0.f, rlo = 0.f, mhi = 0.f, rhi = 0.f;
      if (BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ga[j] = gamma[ch0 + j];
          be[j] = beta[ch0 + j];
        }
        mlo = mean[b * g.G + g0];
        rlo = rstd[b * g.G + g0];
        if (split < 8) {
          mhi = mean[b * g.G + g0 + 1];
          rhi = rstd[b * g.G + g0 + 1];
        }
      }
      float a_lo = 0.f, b_lo = 0.f, a_hi = 0.f, b_hi = 0.f;
      // U rows per trip, every load of a trip issued before the first use (one row per trip compiled to load -> vmcnt(0) ->
      // load -> vmcnt(0) per row).  Same sums, same order; on the step it measured within the noise — eight waves per SIMD
      // already hid those round trips
      constexpr int U = 4;
      for (int rb = r0 + ty; rb < r1; rb += U * g.TY) {
        half8 xu[U], du[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = rb + u * g.TY;
          const long long row = (long long)b * g.HW + (r < r1 ? r : rb);  // (past the slab: re-read the first row, unused)
          xu[u] = *reinterpret_cast<const half8*>(x + row * ldx + ch0);
          if (BWD) du[u] = *reinterpret_cast<const half8*>(dy + row * lddy + ch0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (rb + u * g.TY >= r1) continue;
          const half8 xv = xu[u];
          if (!BWD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float v = (float)xv[j];
              if (j < split) {
                a_lo += v;
                b_lo += v * v;
              } else {
                a_hi += v;
                b_hi += v * v;
              }
            }
          } else {
            const half8 dv = du[u];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const bool lo = j < split;
              float xh = ((float)xv[j] - (lo ? mlo : mhi)) * (lo ? rlo : rhi);
              float d = (float)dv[j];
              if (SILU) {
                float z = xh * ga[j] + be[j];
                float s = vn_sigmoid(z);
                d *= s * (1.f + z * (1.f - s));
              }
              float dxh = d * ga[j];
              if (lo) {
                a_lo += dxh;
                b_lo += dxh * xh;
              } else {
                a_hi += dxh;
                b_hi += dxh * xh;
              }
            }
          }
        }
      }
      vn_fx_add2(&gacc[4 * g0], a_lo, b_lo);
      if (split < 8) vn_fx_add2(&gacc[4 * (g0 + 1)], a_hi, b_hi);
    }
  }
  __syncthreads();
  if constexpr (SLOTS) {
    vn_u64* p = reinterpret_cast<vn_u64*>(part) + ((long long)b * slots + slab % slots) * (4 * g.G);
    for (int i = threadIdx.x; i < 4 * g.G; i += 256) atomicAdd(p + i, gacc[i]);
  } else {
    float* p = part + ((long long)b * g.nslab + slab) * (2 * g.G);
    for (int i = threadIdx.x; i < 2 * g.G; i += 256) p[i] = (float)vn_fx_decode(gacc[2 * i], gacc[2 * i + 1]);
  }
}

// one block per sample: reduce the slab partials (256 threads: 256/G lanes per group, f64 sums).
// FWD: -> mean, rstd.  BWD: -> (S1/n, S2/n).
template <bool BWD>
__global__ __launch_bounds__(256) void gn_finalize_kernel(GNGeom g, const float* __restrict__ part, float eps,
                                                          float* __restrict__ o0, float* __restrict__ o1) {
  __shared__ double red[2][256];
  const int b = blockIdx.x;
  const int lpg = 256 / g.G;  // lanes per group (G <= 64 => lpg >= 4)
  const int gi = threadIdx.x % g.G, ln = threadIdx.x / g.G;
  double s0 = 0.0, s1 = 0.0;
  if (ln < lpg) {
    for (int s = ln; s < g.nslab; s += lpg) {
      const float* p = part + ((long long)b * g.nslab + s) * (2 * g.G) + 2 * gi;
      s0 += (double)p[0];
      s1 += (double)p[1];
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x < g.G) {
    s0 = 0.0;
    s1 = 0.0;
    for (int l = 0; l < lpg; ++l) {
      s0 += red[0][l * g.G + gi];
      s1 += red[1][l * g.G + gi];
    }
    if (!BWD) {
      float m, r;
      vn_mean_rstd(s0, s1, g.inv_n, eps, m, r);
      o0[b * g.G + gi] = m;
      o1[b * g.G + gi] = r;
    } else {
      o0[b * g.G + gi] = (float)(s0 * g.inv_n);
      o1[b * g.G + gi] = (float)(s1 * g.inv_n);
    }
  }
}

// mean / rstd of one (sample, group) from the (sum, sum of squares) slots a GEMM epilogue accumulated
__device__ __forceinline__ void gn_stat_from_sums(const GNGeom& g, const float* sums, int slots, int b, int gi, float eps,
                                                  float& m, float& r) {
  vn_u64 t[4] = {0, 0, 0, 0};  // integer totals over the slots: the same bits whatever order the producers arrived in
  for (int sl = 0; sl < slots; ++sl) {
    const vn_u64* p = reinterpret_cast<const vn_u64*>(sums) + (((long long)b * slots + sl) * g.G + gi) * 4;
#pragma unroll
    for (int w = 0; w < 4; ++w) t[w] += p[w];
  }
  const double s0 = vn_fx_decode(t[0], t[1]), s1 = vn_fx_decode(t[2], t[3]);
  vn_mean_rstd(s0, s1, g.inv_n, eps, m, r);
}

// backward coefficients (S1 / n, S2 / n) of one (sample, group) from slot sums
__device__ __forceinline__ void gn_coef_from_sums(const GNGeom& g, const float* sums, int slots, int b, int gi, float& c1,
                                                  float& c2) {
  vn_u64 t[4] = {0, 0, 0, 0};  // integer totals over the slots: the same bits whatever order the producers arrived in
  for (int sl = 0; sl < slots; ++sl) {
    const vn_u64* p = reinterpret_cast<const vn_u64*>(sums) + (((long long)b * slots + sl) * g.G + gi) * 4;
#pragma unroll
    for (int w = 0; w < 4; ++w) t[w] += p[w];
  }
  const double s0 = vn_fx_decode(t[0], t[1]), s1 = vn_fx_decode(t[2], t[3]);
  c1 = (float)(s0 * g.inv_n);
  c2 = (float)(s1 * g.inv_n);
}

template <bool BWD, bool SILU, bool FROM_SUMS = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(GNGeom g, const half_t* __restrict__ x, long long ldx,
                                                       const half_t* __restrict__ dy, long long lddy,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ c1, const float* __restrict__ c2,
                                                       half_t* __restrict__ out, long long ldo,
                                                       const half_t* __restrict__ accum, long long ldacc,
                                                       const float* __restrict__ sums = nullptr, int slots = 0,
                                                       float eps = 0.f, float* __restrict__ mean_out = nullptr,
                                                       float* __restrict__ rstd_out = nullptr) {
  const int slab = blockIdx.x, b = blockIdx.y;
  const __amdgpu_buffer_rsrc_t rsO = vn_make_rsrc(out, 0x7fffffffu);  // write-through output stores (common.h vn_st16_wt)
  int tx, ty;
  bool active;
  gn_thread_coords(g, tx, ty, active);
  // FROM_SUMS: the statistics pass left fixed-point slot sums; ONE thread per group finishes them for the block (all its
  // slot loads in flight together) and publishes the two numbers through LDS — every thread walking the slots itself was
  // `slots` serialised L2 round trips at the head of each block, more than a small slab's whole streaming time
  __shared__ float s_stat[2][64];
  if constexpr (FROM_SUMS) {
    const int gi = threadIdx.x;
    if (gi < g.G) {
      vn_u64 t[4] = {0, 0, 0, 0};
      const vn_u64* p0 = reinterpret_cast<const vn_u64*>(sums) + ((long long)b * slots * g.G + gi) * 4;
      vn_u64 v[8][4];
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) {
        const vn_u64* p = p0 + (long long)(sl < slots ? sl : 0) * g.G * 4;
#pragma unroll
        for (int w = 0; w < 4; ++w) v[sl][w] = p[w];
      }
#pragma unroll
      for (int sl = 0; sl < 8; ++sl)
#pragma unroll
        for (int w = 0; w < 4; ++w) t[w] += sl < slots ? v[sl][w] : 0;
      for (int s0 = 8; s0 < slots; s0 += 8) {  // (more than eight slots: further batches of eight)
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
          const vn_u64* p = p0 + (long long)(s0 + sl < slots ? s0 + sl : 0) * g.G * 4;
#pragma unroll
          for (int w = 0; w < 4; ++w) v[sl][w] = p[w];
        }
#pragma unroll
        for (int sl = 0; sl < 8; ++sl)
#pragma unroll
          for (int w = 0; w < 4; ++w) t[w] += s0 + sl < slots ? v[sl][w] : 0;
      }
      const double s0 = vn_fx_decode(t[0], t[1]), s1 = vn_fx_decode(t[2], t[3]);
      if constexpr (!BWD) {
        float m, r;
        vn_mean_rstd(s0, s1, g.inv_n, eps, m, r);
        s_stat[0][gi] = m;
        s_stat[1][gi] = r;
        if (slab == 0) {  // published for the backward
          mean_out[b * g.G + gi] = m;
          rstd_out[b * g.G + gi] = r;
        }
      } else {
        s_stat[0][gi] = (float)(s0 * g.inv_n);
        s_stat[1][gi] = (float)(s1 * g.inv_n);
      }
    }
    __syncthreads();
  }
  if (!active) return;
  const int r0 = slab * g.rps;
  const int r1 = min(r0 + g.rps, g.HW);
  for (int cb = 0; cb < g.CC; cb += g.TX) {
    const int cx = cb + tx;
    if (cx >= g.CC) break;
    const int ch0 = cx * 8;
    const int g0 = ch0 / g.cpg;
    const int split = (g0 + 1) * g.cpg - ch0;
    const int g1 = (split < 8) ? g0 + 1 : g0;
    float ga[8], be[8];
    {  // four 16-byte loads, not sixteen scalar ones
      const f32x4 ga0 = *reinterpret_cast<const f32x4*>(gamma + ch0), ga1 = *reinterpret_cast<const f32x4*>(gamma + ch0 + 4);
      const f32x4 be0 = *reinterpret_cast<const f32x4*>(beta + ch0), be1 = *reinterpret_cast<const f32x4*>(beta + ch0 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ga[j] = ga0[j];
        ga[4 + j] = ga1[j];
        be[j] = be0[j];
        be[4 + j] = be1[j];
      }
    }
    float mlo, rlo, mhi, rhi;
    if constexpr (FROM_SUMS && !BWD) {
      mlo = s_stat[0][g0];
      rlo = s_stat[1][g0];
      mhi = s_stat[0][g1];
      rhi = s_stat[1][g1];
    } else {
      mlo = mean[b * g.G + g0];
      rlo = rstd[b * g.G + g0];
      mhi = mean[b * g.G + g1];
      rhi = rstd[b * g.G + g1];
    }
    float c1lo = 0.f, c2lo = 0.f, c1hi = 0.f, c2hi = 0.f;
    if constexpr (BWD && FROM_SUMS) {
      c1lo = s_stat[0][g0];
      c2lo = s_stat[1][g0];
      c1hi = s_stat[0][g1];
      c2hi = s_stat[1][g1];
    } else if (BWD) {
      c1lo = c1[b * g.G + g0];
      c2lo = c2[b * g.G + g0];
      c1hi = c1[b * g.G + g1];
      c2hi = c2[b * g.G + g1];
    }
    int r = r0 + ty;
    float sc[8], sh[8];  // forward: y = x * sc + sh per channel of this thread's chunk
    if (!BWD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool lo = j < split;
        sc[j] = (lo ? rlo : rhi) * ga[j];
        sh[j] = fmaf(-(lo ? mlo : mhi), sc[j], be[j]);
      }
    }
    if (!BWD) {
      // forward: four rows in flight per thread (a single 16-byte load per trip leaves HBM latency exposed:
      // 3.5 TB/s measured on the VAE's 512x512x128 tensors)
      constexpr int U = GN_APPLY_U;
      for (; r + (U - 1) * g.TY < r1; r += U * g.TY) {
        half8 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
          // non-temporal: after this read x is needed again only by the backward, milliseconds later (never, for the VAE) —
          // it must not push the y lines the next convolution is about to read out of the caches (+0.25 % on the step,
          // profiles/r06_gn_nt_load_ab1.txt; the same hint on the backward kernels' operands measured +-0, _ab2)
          xv[u] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(x + ((long long)b * g.HW + r + u * g.TY) * ldx + ch0));
#pragma unroll
        for (int u = 0; u < U; ++u) {
          half8 ov;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float z = fmaf((float)xv[u][j], sc[j], sh[j]);  // (x - mean) * rstd * gamma + beta, folded per channel
            if (SILU) z = vn_silu(z);
            ov[j] = (half_t)z;
          }
          vn_st16_wt(rsO, (uint32_t)((((long long)b * g.HW + r + u * g.TY) * ldo + ch0) * 2), ov);
        }
      }
    }
    for (; r < r1; r += g.TY) {
      const long long row = (long long)b * g.HW + r;
      half8 xv = *reinterpret_cast<const half8*>(x + row * ldx + ch0);
      half8 ov;
      if (!BWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool lo = j < split;
          float xh = ((float)xv[j] - (lo ? mlo : mhi)) * (lo ? rlo : rhi);
          float z = xh * ga[j] + be[j];
          if (SILU) z = vn_silu(z);
          ov[j] = (half_t)z;
        }
      } else {
        half8 dv = *reinterpret_cast<const half8*>(dy + row * lddy + ch0);
        half8 av;
        if (accum) av = *reinterpret_cast<const half8*>(accum + row * ldacc + ch0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool lo = j < split;
          const float rs = lo ? rlo : rhi;
          float xh = ((float)xv[j] - (lo ? mlo : mhi)) * rs;
          float d = (float)dv[j];
          if (SILU) {
            float z = xh * ga[j] + be[j];
            float s = vn_sigmoid(z);
            d *= s * (1.f + z * (1.f - s));
          }
          float dxh = d * ga[j];
          float dx = rs * (dxh - (lo ? c1lo : c1hi) - xh * (lo ? c2lo : c2hi));
          if (accum) dx += (float)av[j];
          ov[j] = (half_t)dx;
        }
      }
      vn_st16_wt(rsO, (uint32_t)((row * ldo + ch0) * 2), ov);
    }
  }
}

// ------------------------------------------------------------------------------------------
// One-launch GroupNorm for small tensors (the low-resolution UNet levels): block (g, b) owns one
// group of one sample, walks it twice -- statistics, then apply; the second walk hits L2 -- instead of the three
// launches above.  At these sizes the three-launch form is launch-latency-bound (the 45 un-fused forward GroupNorms of
// the SD-1.5 step moved 0.42 GB in 0.86 ms), one launch is not.  Elements are addressed as half2 pairs (cpg is even;
// a group's channel range is only 4-byte aligned when cpg = 10).
// ------------------------------------------------------------------------------------------
template <bool BWD, bool SILU>
__global__ __launch_bounds__(512) void gn_small_kernel(int HW, int C, int G, const half_t* __restrict__ x, long long ldx,
                                                       const half_t* __restrict__ dy, long long lddy,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ mean, float* __restrict__ rstd, float eps,
                                                       half_t* __restrict__ out, long long ldo,
                                                       const half_t* __restrict__ accum, long long ldacc, double inv_n) {
  // The slice of one (sample, group) is at most 40 KiB forward / 32 KiB backward (gn_use_small), i.e. at most MAXU
  // half2 pairs per thread: every load of the block is issued up front, the slice stays in registers between the
  // statistics and the apply step (one trip to memory, no second walk), and the row / column of a pair come from a
  // float-reciprocal division instead of an integer one.
  constexpr int MAXU = BWD ? 16 : 20;
  constexpr int NTH = 512;
  __shared__ double red[2][8];
  __shared__ float stat[2];
  __shared__ float sga[128], sbe[128];  // this group's gamma / beta (cpg <= 128)
  const int gi = blockIdx.x, b = blockIdx.y;
  const int cpg = C / G, hp = cpg / 2;  // pairs per row
  const int c0 = gi * cpg;
  const int total = HW * hp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef half2v half2_t;
  for (int i = threadIdx.x; i < cpg; i += NTH) {
    sga[i] = gamma[c0 + i];
    sbe[i] = beta[c0 + i];
  }
  float m = 0.f, rs = 0.f;
  if (BWD) {
    m = mean[b * G + gi];
    rs = rstd[b * G + gi];
  }
  const half_t* xb = x + (long long)b * HW * ldx + c0;
  const half_t* dyb = BWD ? dy + (long long)b * HW * lddy + c0 : nullptr;
  const half_t* ab = accum ? accum + (long long)b * HW * ldacc + c0 : nullptr;
  const float rcp_hp = __builtin_amdgcn_rcpf((float)hp);
  half2_t xv[MAXU], dv[BWD ? MAXU : 1], av[BWD ? MAXU : 1];
  int rows[MAXU], cps[MAXU];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int p = threadIdx.x + u * NTH;
    int cp;
    rows[u] = vn_divmod(p, hp, rcp_hp, cp);
    cps[u] = cp * 2;
    xv[u] = half2_t{(half_t)0.f, (half_t)0.f};
    if (BWD) {
      dv[u] = xv[u];
      av[u] = xv[u];
    }
    if (p < total) {
      xv[u] = *reinterpret_cast<const half2_t*>(xb + (long long)rows[u] * ldx + cps[u]);
      if (BWD) {
        dv[u] = *reinterpret_cast<const half2_t*>(dyb + (long long)rows[u] * lddy + cps[u]);
        if (ab) av[u] = *reinterpret_cast<const half2_t*>(ab + (long long)rows[u] * ldacc + cps[u]);
      }
    }
  }
  __syncthreads();  // sga / sbe
  // ---- statistics: FWD (sum, sum of squares); BWD (sum dxh, sum dxh * xhat); pairs past the end hold zeros
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    if (!BWD) {
      const float v0 = (float)xv[u][0], v1 = (float)xv[u][1];
      s0 += v0 + v1;
      s1 += v0 * v0 + v1 * v1;
    } else if (threadIdx.x + u * NTH < total) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float ga = sga[cps[u] + j];
        const float xh = ((float)xv[u][j] - m) * rs;
        float d = (float)dv[u][j];
        if (SILU) {
          const float z = xh * ga + sbe[cps[u] + j];
          const float sg = vn_sigmoid(z);
          d *= sg * (1.f + z * (1.f - sg));
        }
        const float dxh = d * ga;
        s0 += dxh;
        s1 += dxh * xh;
      }
    }
  }
  double d0 = (double)s0, d1 = (double)s1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    d0 += __shfl_xor(d0, off);
    d1 += __shfl_xor(d1, off);
  }
  if (lane == 0) {
    red[0][wave] = d0;
    red[1][wave] = d1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t0 = 0.0, t1 = 0.0;
    for (int w = 0; w < (NTH >> 6); ++w) {
      t0 += red[0][w];
      t1 += red[1][w];
    }
    if (!BWD) {
      vn_mean_rstd(t0, t1, inv_n, eps, stat[0], stat[1]);
      mean[b * G + gi] = stat[0];
      rstd[b * G + gi] = stat[1];
    } else {
      stat[0] = (float)(t0 * inv_n);
      stat[1] = (float)(t1 * inv_n);
    }
  }
  __syncthreads();
  const float a0 = stat[0], a1 = stat[1];  // FWD: mean, rstd;  BWD: c1, c2
  // ---- apply, from the registers
  half_t* ob = out + (long long)b * HW * ldo + c0;
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    if (threadIdx.x + u * NTH >= total) break;
    half2_t ov;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float ga = sga[cps[u] + j], be = sbe[cps[u] + j];
      if (!BWD) {
        float z = ((float)xv[u][j] - a0) * a1 * ga + be;
        if (SILU) z = vn_silu(z);
        ov[j] = (half_t)z;
      } else {
        const float xh = ((float)xv[u][j] - m) * rs;
        float d = (float)dv[u][j];
        if (SILU) {
          const float z = xh * ga + be;
          const float sg = vn_sigmoid(z);
          d *= sg * (1.f + z * (1.f - sg));
        }
        float dx = rs * (d * ga - a0 - xh * a1);
        if (ab) dx += (float)av[u][j];
        ov[j] = (half_t)dx;
      }
    }
    *reinterpret_cast<half2_t*>(ob + (long long)rows[u] * ldo + cps[u]) = ov;
  }
}

// one launch where it measures faster than three (tools/lab/gn_paths.py, bs=4: slices <= 40 KiB forward, <= 32 KiB
// backward -- the 8x8 and 16x16 levels and the narrow 32x32 layers; 128 blocks cannot stream the bigger ones fast enough)
inline bool gn_use_small(int Bn, int HW, int C, int G, bool bwd) {
  const int cpg = C / G;
  if (cpg % 2 != 0 || cpg > 128 || getenv("VNETI_GN_NO_SMALL")) return false;
  return (long long)HW * cpg * 2 <= (bwd ? 32 : 40) * 1024 && Bn * G >= 64;
}

int gn_geom(GNGeom& g, int Bn, int HW, int C, int G) {
  if (Bn <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || C % G != 0 || C % 8 != 0) return -1;
  g.Bn = Bn;
  g.HW = HW;
  g.C = C;
  g.G = G;
  g.cpg = C / G;
  if (g.cpg < 4) return -1;
  g.CC = C / 8;
  g.TX = g.CC < 256 ? g.CC : 256;
  g.TY = 256 / g.TX;
  int rps = 8192 / C;
  if (rps < 1) rps = 1;
  rps = ((rps + g.TY - 1) / g.TY) * g.TY;
  int nslab = (HW + rps - 1) / rps;
  while (nslab > 256) {
    rps *= 2;
    nslab = (HW + rps - 1) / rps;
  }
  g.rps = rps;
  g.nslab = nslab;
  g.inv_n = 1.0 / ((double)HW * g.cpg);
  return 0;
}

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, up to 4 chunks of 8 elements per lane (C <= 2048).
// ------------------------------------------------------------------------------------------
template <bool XF32>
__device__ __forceinline__ void ln_load(const void* x, long long off, float v[8]) {
  if (XF32) {
    const float* p = reinterpret_cast<const float*>(x) + off;
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = a[j];
      v[4 + j] = b[j];
    }
  } else {
    half8 h = *reinterpret_cast<const half8*>(reinterpret_cast<const half_t*>(x) + off);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)h[j];
  }
}
template <bool F32>
__device__ __forceinline__ void ln_store(void* y, long long off, const float v[8]) {
  // write-through (common.h vn_st16_wt): `y` is the kernel argument, wave-uniform
  const __amdgpu_buffer_rsrc_t rs = vn_make_rsrc(y, 0x7fffffffu);
  if (F32) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    vn_st16_wt(rs, (uint32_t)(off * 4), a);
    vn_st16_wt(rs, (uint32_t)(off * 4 + 16), b);
  } else {
    half8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (half_t)v[j];
    vn_st16_wt(rs, (uint32_t)(off * 2), h);
  }
}

constexpr int LN_MAXC = 4;

// Both kernels are latency-, not bandwidth-bound (2.6 .. 10 MB per launch, 140 launches per step): every load is issued up
// front and unconditionally -- a lane whose chunk index is past the row re-reads the last chunk and is masked out of the
// sums and stores -- and gamma / beta come as two 16-byte loads per chunk.  With per-chunk `if (c < CC)` blocks and
// scalar gamma loads the compiler emitted 12 serialised load -> s_waitcnt vmcnt(0) round trips in the backward kernel.
// NC = chunks of 8 per lane that the row length needs (1 for C <= 512 ... 4 for C <= 2048).
__device__ __forceinline__ void ln_load_f32x8(const float* p, float v[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = a[j];
    v[4 + j] = b[j];
  }
}

template <bool XF32, int NC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, long long ldx,
                                                     half_t* __restrict__ y, long long ldy,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int CC = C / 8;
  float v[NC][8], ga[NC][8], be[NC][8];
  bool ok[NC];
  int cc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    ok[i] = c < CC;
    cc[i] = ok[i] ? c : CC - 1;
    ln_load<XF32>(x, (long long)row * ldx + cc[i] * 8, v[i]);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    ln_load_f32x8(gamma + cc[i] * 8, ga[i]);
    ln_load_f32x8(beta + cc[i] * 8, be[i]);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += v[i][j];
    s += ok[i] ? t : 0.f;
  }
  s = wave_sum(s);
  const float inv_c = __builtin_amdgcn_rcpf((float)C);  // (1 ulp; an IEEE division is ~10 issues on the critical path of every row)
  const float m = s * inv_c;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[i][j] - m;
      t += d * d;
    }
    q += ok[i] ? t : 0.f;
  }
  q = wave_sum(q);
  const float rs = rsqrtf(q * inv_c + eps);
  if (lane == 0) {
    if (mean) mean[row] = m;
    if (rstd) rstd[row] = rs;
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (ok[i]) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - m) * rs * ga[i][j] + be[i][j];
      ln_store<false>(y, (long long)row * ldy + cc[i] * 8, o);
    }
  }
}

template <bool DYF32, bool XF32, bool DXF32, int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, long long lddy,
                                                     const void* __restrict__ x, long long ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, void* __restrict__ dx,
                                                     long long lddx, const void* __restrict__ accum,
                                                     long long ldacc, half_t* __restrict__ dx16, long long ld16,
                                                     int rows, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int CC = C / 8;
  float xh[NC][8], dh[NC][8], ac[NC][8];
  bool ok[NC];
  int cc[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    ok[i] = c < CC;
    cc[i] = ok[i] ? c : CC - 1;
    ln_load<XF32>(x, (long long)row * ldx + cc[i] * 8, xh[i]);
    ln_load<DYF32>(dy, (long long)row * lddy + cc[i] * 8, dh[i]);
  }
  const float m = mean[row], rs = rstd[row];
  if (accum) {  // kernel-uniform; the accumulated gradient is requested together with the rest
#pragma unroll
    for (int i = 0; i < NC; ++i) ln_load<DXF32>(accum, (long long)row * ldacc + cc[i] * 8, ac[i]);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    float g8[8];
    ln_load_f32x8(gamma + cc[i] * 8, g8);
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[i][j] = (xh[i][j] - m) * rs;
      dh[i][j] = dh[i][j] * g8[j];
      t1 += dh[i][j];
      t2 += dh[i][j] * xh[i][j];
    }
    s1 += ok[i] ? t1 : 0.f;
    s2 += ok[i] ? t2 : 0.f;
  }
  const float inv_c = __builtin_amdgcn_rcpf((float)C);
  s1 = wave_sum(s1) * inv_c;
  s2 = wave_sum(s2) * inv_c;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    if (ok[i]) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rs * (dh[i][j] - s1 - xh[i][j] * s2);
      if (accum) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += ac[i][j];
      }
      ln_store<DXF32>(dx, (long long)row * lddx + cc[i] * 8, o);
      if (dx16) ln_store<false>(dx16, (long long)row * ld16 + cc[i] * 8, o);  // f16 copy = next GEMM's operand
    }
  }
}

// ------------------------------------------------------------------------------------------
// in-place row softmax on f16 (VAE mid-block attention, N = H*W keys, single head d=512)
// ------------------------------------------------------------------------------------------
constexpr int SM_MAXC = 8;
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ x, long long ld, int cols) {
  __shared__ float red[4];
  half_t* p = x + (long long)blockIdx.x * ld;
  const int CC = cols / 8;
  float v[SM_MAXC][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    int c = threadIdx.x + 256 * i;
    if (c < CC) {
      half8 h = *reinterpret_cast<const half8*>(p + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[i][j] = (float)h[j];
        mx = fmaxf(mx, v[i][j]);
      }
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    int c = threadIdx.x + 256 * i;
    if (c < CC) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[i][j] = __expf(v[i][j] - mx);
        s += v[i][j];
      }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  s = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    int c = threadIdx.x + 256 * i;
    if (c < CC) {
      half8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (half_t)(v[i][j] * inv);
      *reinterpret_cast<half8*>(p + c * 8) = h;
    }
  }
}

}  // namespace

extern "C" long long vneti_groupnorm_ws_floats(int Bn, int HW, int C, int G) {
  GNGeom g;
  if (gn_geom(g, Bn, HW, C, G) != 0) return -1;
  // slab partials + (c1, c2) for the backward
  return (long long)Bn * g.nslab * 2 * G + 2LL * Bn * G;
}

extern "C" int vneti_groupnorm_fwd(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                                   const float* beta, float* mean, float* rstd, float* ws, int Bn, int HW,
                                   int C, int G, float eps, int silu, void* stream) {
  GNGeom g;
  VN_REQUIRE(gn_geom(g, Bn, HW, C, G) == 0, "groupnorm: unsupported shape B=%d HW=%d C=%d G=%d", Bn, HW, C, G);
  VN_REQUIRE(x && y && gamma && beta && mean && rstd && ws, "groupnorm_fwd: null pointer");
  VN_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "groupnorm_fwd: ld must be a multiple of 8");
  hipStream_t st = (hipStream_t)stream;
  if (gn_use_small(Bn, HW, C, G, false)) {
    if (silu)
      hipLaunchKernelGGL((gn_small_kernel<false, true>), dim3(G, Bn), dim3(512), 0, st, HW, C, G, (const half_t*)x, ldx,
                         (const half_t*)nullptr, 0LL, gamma, beta, mean, rstd, eps, (half_t*)y, ldy,
                         (const half_t*)nullptr, 0LL, g.inv_n);
    else
      hipLaunchKernelGGL((gn_small_kernel<false, false>), dim3(G, Bn), dim3(512), 0, st, HW, C, G, (const half_t*)x, ldx,
                         (const half_t*)nullptr, 0LL, gamma, beta, mean, rstd, eps, (half_t*)y, ldy,
                         (const half_t*)nullptr, 0LL, g.inv_n);
    return vneti_check_launch("groupnorm_fwd");
  }
  dim3 grid(g.nslab, Bn);
  hipLaunchKernelGGL((gn_stats_kernel<false, false>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                     (const half_t*)nullptr, 0LL, gamma, beta, (const float*)nullptr, (const float*)nullptr, ws);
  hipLaunchKernelGGL((gn_finalize_kernel<false>), dim3(Bn), dim3(256), 0, st, g, (const float*)ws, eps, mean, rstd);
  if (silu)
    hipLaunchKernelGGL((gn_apply_kernel<false, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)nullptr, 0LL, gamma, beta, (const float*)mean, (const float*)rstd,
                       (const float*)nullptr, (const float*)nullptr, (half_t*)y, ldy, (const half_t*)nullptr, 0LL);
  else
    hipLaunchKernelGGL((gn_apply_kernel<false, false>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)nullptr, 0LL, gamma, beta, (const float*)mean, (const float*)rstd,
                       (const float*)nullptr, (const float*)nullptr, (half_t*)y, ldy, (const half_t*)nullptr, 0LL);
  return vneti_check_launch("groupnorm_fwd");
}

extern "C" int vneti_groupnorm_fwd_sums(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                                        const float* beta, const void* sums_v, int slots, float* mean, float* rstd,
                                        int Bn, int HW, int C, int G, float eps, int silu, void* stream) {
  const float* sums = reinterpret_cast<const float*>(sums_v);  // 64-bit fixed-point words, see common.h (typed float* internally)
  GNGeom g;
  VN_REQUIRE(gn_geom(g, Bn, HW, C, G) == 0, "groupnorm: unsupported shape B=%d HW=%d C=%d G=%d", Bn, HW, C, G);
  VN_REQUIRE(x && y && gamma && beta && sums && mean && rstd && slots > 0, "groupnorm_fwd_sums: null pointer");
  VN_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "groupnorm_fwd_sums: ld % 8 != 0");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(g.nslab, Bn);
  if (silu)
    hipLaunchKernelGGL((gn_apply_kernel<false, true, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)nullptr, 0LL, gamma, beta, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (half_t*)y, ldy, (const half_t*)nullptr, 0LL,
                       sums, slots, eps, mean, rstd);
  else
    hipLaunchKernelGGL((gn_apply_kernel<false, false, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)nullptr, 0LL, gamma, beta, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (half_t*)y, ldy, (const half_t*)nullptr, 0LL,
                       sums, slots, eps, mean, rstd);
  return vneti_check_launch("groupnorm_fwd_sums");
}

// Two launches instead of three for the tensors that are too big for the one-block-per-group kernel: the statistics go
// straight into caller-zeroed slot sums and the apply kernel finishes them (no finalize launch).  Small tensors take the
// same one-launch kernel as vneti_groupnorm_fwd / _bwd (sums untouched).
extern "C" int vneti_groupnorm_fwd_2l(const void* x, long long ldx, void* y, long long ldy, const float* gamma,
                                      const float* beta, void* sums_v, int slots, float* mean, float* rstd, int Bn, int HW,
                                      int C, int G, float eps, int silu, void* stream) {
  float* sums = reinterpret_cast<float*>(sums_v);  // 64-bit fixed-point words, see common.h (typed float* internally)
  GNGeom g;
  VN_REQUIRE(gn_geom(g, Bn, HW, C, G) == 0, "groupnorm: unsupported shape B=%d HW=%d C=%d G=%d", Bn, HW, C, G);
  VN_REQUIRE(x && y && gamma && beta && mean && rstd && sums && slots > 0, "groupnorm_fwd_2l: null pointer");
  VN_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "groupnorm_fwd_2l: ld must be a multiple of 8");
  if (gn_use_small(Bn, HW, C, G, false))
    return vneti_groupnorm_fwd(x, ldx, y, ldy, gamma, beta, mean, rstd, sums, Bn, HW, C, G, eps, silu, stream);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(g.nslab, Bn);
  hipLaunchKernelGGL((gn_stats_kernel<false, false, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                     (const half_t*)nullptr, 0LL, gamma, beta, (const float*)nullptr, (const float*)nullptr, sums, slots);
  return vneti_groupnorm_fwd_sums(x, ldx, y, ldy, gamma, beta, sums, slots, mean, rstd, Bn, HW, C, G, eps, silu, stream);
}

extern "C" int vneti_groupnorm_bwd_2l(const void* dy, long long lddy, const void* x, long long ldx, const float* gamma,
                                      const float* beta, const float* mean, const float* rstd, void* dx, long long lddx,
                                      const void* dx_accum, long long ldacc, void* sums_v, int slots, float* ws, int Bn,
                                      int HW, int C, int G, int silu, void* stream);

extern "C" int vneti_groupnorm_bwd(const void* dy, long long lddy, const void* x, long long ldx,
                                   const float* gamma, const float* beta, const float* mean, const float* rstd,
                                   void* dx, long long lddx, const void* dx_accum, long long ldacc, float* ws,
                                   int Bn, int HW, int C, int G, int silu, void* stream) {
  GNGeom g;
  VN_REQUIRE(gn_geom(g, Bn, HW, C, G) == 0, "groupnorm: unsupported shape B=%d HW=%d C=%d G=%d", Bn, HW, C, G);
  VN_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && ws, "groupnorm_bwd: null pointer");
  VN_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldacc % 8 == 0, "groupnorm_bwd: ld % 8 != 0");
  hipStream_t st = (hipStream_t)stream;
  if (gn_use_small(Bn, HW, C, G, true)) {
    if (silu)
      hipLaunchKernelGGL((gn_small_kernel<true, true>), dim3(G, Bn), dim3(512), 0, st, HW, C, G, (const half_t*)x, ldx,
                         (const half_t*)dy, lddy, gamma, beta, const_cast<float*>(mean), const_cast<float*>(rstd), 0.f,
                         (half_t*)dx, lddx, (const half_t*)dx_accum, ldacc, g.inv_n);
    else
      hipLaunchKernelGGL((gn_small_kernel<true, false>), dim3(G, Bn), dim3(512), 0, st, HW, C, G, (const half_t*)x, ldx,
                         (const half_t*)dy, lddy, gamma, beta, const_cast<float*>(mean), const_cast<float*>(rstd), 0.f,
                         (half_t*)dx, lddx, (const half_t*)dx_accum, ldacc, g.inv_n);
    return vneti_check_launch("groupnorm_bwd");
  }
  dim3 grid(g.nslab, Bn);
  float* c1 = ws + (long long)Bn * g.nslab * 2 * G;
  float* c2 = c1 + (long long)Bn * G;
  if (silu)
    hipLaunchKernelGGL((gn_stats_kernel<true, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)dy, lddy, gamma, beta, mean, rstd, ws);
  else
    hipLaunchKernelGGL((gn_stats_kernel<true, false>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)dy, lddy, gamma, beta, mean, rstd, ws);
  hipLaunchKernelGGL((gn_finalize_kernel<true>), dim3(Bn), dim3(256), 0, st, g, (const float*)ws, 0.f, c1, c2);
  if (silu)
    hipLaunchKernelGGL((gn_apply_kernel<true, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)dy, lddy, gamma, beta, mean, rstd, (const float*)c1, (const float*)c2,
                       (half_t*)dx, lddx, (const half_t*)dx_accum, ldacc);
  else
    hipLaunchKernelGGL((gn_apply_kernel<true, false>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx,
                       (const half_t*)dy, lddy, gamma, beta, mean, rstd, (const float*)c1, (const float*)c2,
                       (half_t*)dx, lddx, (const half_t*)dx_accum, ldacc);
  return vneti_check_launch("groupnorm_bwd");
}

extern "C" int vneti_groupnorm_bwd_2l(const void* dy, long long lddy, const void* x, long long ldx, const float* gamma,
                                      const float* beta, const float* mean, const float* rstd, void* dx, long long lddx,
                                      const void* dx_accum, long long ldacc, void* sums_v, int slots, float* ws, int Bn,
                                      int HW, int C, int G, int silu, void* stream) {
  float* sums = reinterpret_cast<float*>(sums_v);  // 64-bit fixed-point words, see common.h (typed float* internally)
  GNGeom g;
  VN_REQUIRE(gn_geom(g, Bn, HW, C, G) == 0, "groupnorm: unsupported shape B=%d HW=%d C=%d G=%d", Bn, HW, C, G);
  VN_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && sums && slots > 0, "groupnorm_bwd_2l: null pointer");
  VN_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldacc % 8 == 0, "groupnorm_bwd_2l: ld % 8 != 0");
  if (gn_use_small(Bn, HW, C, G, true))
    return vneti_groupnorm_bwd(dy, lddy, x, ldx, gamma, beta, mean, rstd, dx, lddx, dx_accum, ldacc, ws, Bn, HW, C, G, silu,
                               stream);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(g.nslab, Bn);
#define GN_BWD_2L(S)                                                                                                     \
  hipLaunchKernelGGL((gn_stats_kernel<true, S, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx, (const half_t*)dy, \
                     lddy, gamma, beta, mean, rstd, sums, slots);                                                         \
  hipLaunchKernelGGL((gn_apply_kernel<true, S, true>), grid, dim3(256), 0, st, g, (const half_t*)x, ldx, (const half_t*)dy, \
                     lddy, gamma, beta, mean, rstd, (const float*)nullptr, (const float*)nullptr, (half_t*)dx, lddx,      \
                     (const half_t*)dx_accum, ldacc, (const float*)sums, slots, 0.f, (float*)nullptr, (float*)nullptr)
  if (silu) {
    GN_BWD_2L(true);
  } else {
    GN_BWD_2L(false);
  }
#undef GN_BWD_2L
  return vneti_check_launch("groupnorm_bwd_2l");
}

extern "C" int vneti_layernorm_fwd(const void* x, int x_is_f32, long long ldx, void* y, long long ldy,
                                   const float* gamma, const float* beta, float* mean, float* rstd, int rows,
                                   int C, float eps, void* stream) {
  VN_REQUIRE(x && y && gamma && beta, "layernorm_fwd: null pointer");
  VN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * LN_MAXC, "layernorm: unsupported C=%d", C);
  VN_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "layernorm_fwd: ld % 8 != 0");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4));
#define LN_FWD(F, NCH)                                                                                        \
  hipLaunchKernelGGL((ln_fwd_kernel<F, NCH>), grid, dim3(256), 0, st, x, ldx, (half_t*)y, ldy, gamma, beta, mean, \
                     rstd, rows, C, eps)
  const int nc = cdiv(C / 8, 64);
  if (x_is_f32) {
    switch (nc) {
      case 1: LN_FWD(true, 1); break;
      case 2: LN_FWD(true, 2); break;
      case 3: LN_FWD(true, 3); break;
      default: LN_FWD(true, 4); break;
    }
  } else {
    switch (nc) {
      case 1: LN_FWD(false, 1); break;
      case 2: LN_FWD(false, 2); break;
      case 3: LN_FWD(false, 3); break;
      default: LN_FWD(false, 4); break;
    }
  }
#undef LN_FWD
  return vneti_check_launch("layernorm_fwd");
}

extern "C" int vneti_layernorm_bwd(const void* dy, int dy_is_f32, long long lddy, const void* x, int x_is_f32,
                                   long long ldx, const float* gamma, const float* mean, const float* rstd,
                                   void* dx, int dx_is_f32, long long lddx, const void* dx_accum, long long ldacc,
                                   void* dx_f16_copy, long long ldcopy, int rows, int C, void* stream) {
  VN_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  VN_REQUIRE(ldcopy % 8 == 0, "layernorm_bwd: ldcopy % 8 != 0");
  VN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * LN_MAXC, "layernorm: unsupported C=%d", C);
  VN_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && ldacc % 8 == 0, "layernorm_bwd: ld % 8 != 0");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(rows, 4));
  const int nc = cdiv(C / 8, 64);
#define LN_BWD_N(A, B, Cc, NCH)                                                                                         \
  hipLaunchKernelGGL((ln_bwd_kernel<A, B, Cc, NCH>), grid, dim3(256), 0, st, dy, lddy, x, ldx, gamma, mean, rstd, dx, \
                     lddx, dx_accum, ldacc, (half_t*)dx_f16_copy, ldcopy, rows, C)
#define LN_BWD(A, B, Cc)                      \
  switch (nc) {                               \
    case 1: LN_BWD_N(A, B, Cc, 1); break;     \
    case 2: LN_BWD_N(A, B, Cc, 2); break;     \
    case 3: LN_BWD_N(A, B, Cc, 3); break;     \
    default: LN_BWD_N(A, B, Cc, 4); break;    \
  }
  int key = (dy_is_f32 ? 4 : 0) | (x_is_f32 ? 2 : 0) | (dx_is_f32 ? 1 : 0);
  switch (key) {
    case 0: LN_BWD(false, false, false); break;
    case 1: LN_BWD(false, false, true); break;
    case 2: LN_BWD(false, true, false); break;
    case 3: LN_BWD(false, true, true); break;
    case 4: LN_BWD(true, false, false); break;
    case 5: LN_BWD(true, false, true); break;
    case 6: LN_BWD(true, true, false); break;
    default: LN_BWD(true, true, true); break;
  }
#undef LN_BWD
#undef LN_BWD_N
  return vneti_check_launch("layernorm_bwd");
}

extern "C" int vneti_softmax_rows_f16(void* x, long long ld, int rows, int cols, void* stream) {
  VN_REQUIRE(x && rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 8 * 256 * SM_MAXC && ld % 8 == 0,
             "softmax_rows: unsupported cols=%d ld=%lld", cols, ld);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (half_t*)x, ld, cols);
  return vneti_check_launch("softmax_rows");
}
