// NeTI text-path kernels: the fused mapper (the only trainable network of the train step),
// the placeholder-overwriting embedding lookup, and the textual-bypass + final LayerNorm pair.
//
// Reference code restated here (see oracle/sd_ref.py for the line-by-line CPU version):
//   models/neti_mapper.py:165-197,368-438,542-578  NeTIMapper.forward (arch_view_net = 15)
//   models/positional_encoding.py:174-195          FourierPositionalEncodingNDims.forward
//   models/net_clip_text_embedding.py:34-137       token embed -> overwrite placeholder rows -> + position
//   models/neti_clip_text_encoder.py:121-185       bypass injection on a clone + final_layer_norm x2
// The reference runs these once per UNet cross-attention layer (16 python passes with ~100 host
// syncs, training/coach.py:289-305); here all (layer, sample) pairs are rows of one launch.
#include "common.h"
#include "../../include/vneti.h"

namespace {

constexpr int MAXH = 192;  // max hidden width / encoding width of the mapper MLP (legacy path: 10 x 16 = 160 anchors)

struct MapperParams {
  // offsets (in floats) into the flat parameter / gradient bucket, state_dict order:
  // net.0.weight [hd][E], net.0.bias, net.1.weight, net.1.bias, net.3.weight [hd][hd], net.3.bias,
  // net.4.weight, net.4.bias, output_layer.0.weight [OD][hd], output_layer.0.bias [OD]
  int w0, b0, g1, be1, w3, b3, g2, be2, wo, bo;
  int E, hd, OD, D;  // E = encoding dim (64), OD = 2*D with bypass else D
};

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.01f * x; }

// One block of MT threads per mapper row r = (layer, sample).  The weights are the only sizeable operand (0.79 MB for the
// 1536 x 128 output layer) and every row's block reads all of them out of L2, so the row kernels are latency problems:
// 16 waves per block, every weight read coalesced along the contiguous index (16-byte loads when the bucket allows: VEC).
constexpr int MT = 1024;

// all-reduce over the 16 lanes of a DPP row with rotations (four VALU instructions, no LDS crossbar)
__device__ __forceinline__ float group16_sum(float s) {
  s += vn_row_ror<8>(s);
  s += vn_row_ror<4>(s);
  s += vn_row_ror<2>(s);
  s += vn_row_ror<1>(s);
  return s;
}

// sum over the block of one float per thread; `red` holds MT / 64 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();  // red may still be read from an earlier call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < MT / 64; ++i) t += red[i];
  return t;
}

// emit(o, sum_i W[o][i] x[i]) for o < NO (W row-major [NO][NI], x in LDS): 16 lanes per output row, coalesced along i,
// four rows per group and pass so that four L2 round trips are in flight (a pass is load -> dot -> reduce, one chain);
// emit runs in the first lane of the row's group.
template <bool VEC, class F>
__device__ __forceinline__ void matvec_rows(const float* __restrict__ W, const float* x, int NO, int NI, F&& emit) {
  constexpr int U = 4, ROWS = MT / 16;
  const int g = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (int o0 = 0; o0 < NO; o0 += U * ROWS) {
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int o = o0 + u * ROWS + grp;
      s[u] = 0.f;
      if (o < NO) {
        const float* wr = W + (long long)o * NI;
        if constexpr (VEC) {
          for (int i = g * 4; i < NI; i += 64) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wr + i);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
            s[u] += w[0] * v[0] + w[1] * v[1] + w[2] * v[2] + w[3] * v[3];
          }
        } else {
          for (int i = g; i < NI; i += 16) s[u] += wr[i] * x[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int o = o0 + u * ROWS + grp;
      const float t = group16_sum(s[u]);
      if (g == 0 && o < NO) emit(o, t);
    }
  }
}

// out[i] = sum_o W[o][i] x[o] for i < NI (the transposed product; W row-major [NO][NI], x in LDS, out in LDS): the threads
// split into partitions over o, each reading whole rows coalesced along i; `part` holds MT * 4 floats.  Ends with a barrier.
template <bool VEC>
__device__ __forceinline__ void matvec_cols(const float* __restrict__ W, const float* x, int NO, int NI, float* part,
                                            float* out) {
  const int tid = threadIdx.x;
  if constexpr (VEC) {
    const int nq = NI / 4, q = tid % nq, p = tid / nq, np = MT / nq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (p < np)
      for (int o = p; o < NO; o += np) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(W + (long long)o * NI + q * 4);
        const float xv = x[o];
        acc[0] += w[0] * xv;
        acc[1] += w[1] * xv;
        acc[2] += w[2] * xv;
        acc[3] += w[3] * xv;
      }
    if (p < np) *reinterpret_cast<f32x4*>(part + (p * nq + q) * 4) = acc;
    __syncthreads();
    if (tid < NI) {
      float t = 0.f;
      for (int k = 0; k < np; ++k) t += part[k * NI + tid];
      out[tid] = t;
    }
  } else {
    const int j = tid % NI, p = tid / NI, np = MT / NI;
    float acc = 0.f;
    if (p < np)
      for (int o = p; o < NO; o += np) acc += W[(long long)o * NI + j] * x[o];
    if (p < np) part[p * NI + j] = acc;
    __syncthreads();
    if (tid < NI) {
      float t = 0.f;
      for (int k = 0; k < np; ++k) t += part[k * NI + tid];
      out[tid] = t;
    }
  }
  __syncthreads();
}

__device__ void block_layernorm(float* z, const float* gamma, const float* beta, float* xh, float* y, int hd,
                                float* stat) {
  // z[hd] in LDS -> xh (normalised), y = xh*gamma+beta; stat[1]=rstd.  eps = 1e-5 (nn.LayerNorm default)
  if (threadIdx.x < 64) {
    float m = 0.f;
    for (int i = threadIdx.x; i < hd; i += 64) m += z[i];
    m = wave_sum(m) / hd;
    float v = 0.f;
    for (int i = threadIdx.x; i < hd; i += 64) v += (z[i] - m) * (z[i] - m);
    v = wave_sum(v);
    if (threadIdx.x == 0) {
      stat[0] = m;
      stat[1] = rsqrtf(v / hd + 1e-5f);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < hd) {
    float h = (z[threadIdx.x] - stat[0]) * stat[1];
    xh[threadIdx.x] = h;
    y[threadIdx.x] = h * gamma[threadIdx.x] + beta[threadIdx.x];
  }
  __syncthreads();
}

template <bool VEC>
__global__ __launch_bounds__(MT) void mapper_fwd_kernel(MapperParams mp, const float* __restrict__ params,
                                                        const int* __restrict__ slot, long long slot_stride,
                                                        const float* __restrict__ data, int nfeat,
                                                        const float* __restrict__ w_enc,
                                                        const float* __restrict__ hmask, float norm_scale,
                                                        float* __restrict__ word, float* __restrict__ bypass,
                                                        float* __restrict__ save,
                                                        const float* __restrict__ enc_in) {
  __shared__ __attribute__((aligned(16))) float outs[2048 + 64], enc[MAXH], z[MAXH], xh[MAXH], y[MAXH], a[MAXH];
  __shared__ float stat[2], red[MT / 64];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (slot) params += (long long)slot[0] * slot_stride;  // which mapper of a multi-mapper bucket (device-side)
  const int E = mp.E, hd = mp.hd, D = mp.D, OD = mp.OD;
  // per-row save area: enc[E] | xh1[hd] | a1[hd] | xh2[hd] | a2m[hd] | rstd1, rstd2, wnorm, pad
  float* sv = save + (long long)r * (E + 4 * hd + 4);
  if (enc_in) {  // legacy mapper: the first-layer input is the output of its trainable input_layer (computed upstream)
    if (tid < E) enc[tid] = enc_in[(long long)r * E + tid];
  } else if (tid < E / 2) {
    float p = 0.f;
    for (int f = 0; f < nfeat; ++f) p += w_enc[tid * nfeat + f] * data[r * nfeat + f];
    enc[tid] = sinf(p);
    enc[E / 2 + tid] = cosf(p);
  }
  __syncthreads();
  if (tid < E) sv[tid] = enc[tid];
  matvec_rows<VEC>(params + mp.w0, enc, hd, E, [&](int o, float s) { z[o] = s + params[mp.b0 + o]; });
  __syncthreads();
  block_layernorm(z, params + mp.g1, params + mp.be1, xh, y, hd, stat);
  if (tid < hd) {
    a[tid] = leaky(y[tid]);
    sv[E + tid] = xh[tid];
    sv[E + hd + tid] = a[tid];
  }
  if (tid == 0) sv[E + 4 * hd] = stat[1];
  __syncthreads();
  matvec_rows<VEC>(params + mp.w3, a, hd, hd, [&](int o, float s) { z[o] = s + params[mp.b3 + o]; });
  __syncthreads();
  block_layernorm(z, params + mp.g2, params + mp.be2, xh, y, hd, stat);
  if (tid < hd) {
    float v = leaky(y[tid]);
    if (hmask) v *= hmask[r * hd + tid];
    a[tid] = v;
    sv[E + 2 * hd + tid] = xh[tid];
    sv[E + 3 * hd + tid] = v;
  }
  if (tid == 0) sv[E + 4 * hd + 1] = stat[1];
  __syncthreads();
  // output layer: [word | bypass] rows of net.output_layer
  float sq = 0.f;
  matvec_rows<VEC>(params + mp.wo, a, OD, hd, [&](int o, float s) {
    s += params[mp.bo + o];
    outs[o] = s;
    if (o < D) sq += s * s;
  });
  const float nrm = sqrtf(block_sum(sq, red));  // (its barriers also publish outs)
  if (tid == 0) sv[E + 4 * hd + 2] = nrm;
  const float f = norm_scale > 0.f ? norm_scale / fmaxf(nrm, 1e-12f) : 1.f;  // F.normalize(eps=1e-12) * norm_scale
  for (int o = tid; o < OD; o += MT) {
    if (o < D) word[(long long)r * D + o] = outs[o] * f;
    else bypass[(long long)r * D + (o - D)] = outs[o];
  }
}

// backward stage 1: per row, from (d_word, d_bypass) down to the pre-LayerNorm gradients.
// rowgrads layout per row: dout[OD] | dz2[hd] | dy2[hd] | dz1[hd] | dy1[hd]
template <bool VEC>
__global__ __launch_bounds__(MT) void mapper_bwd_rows_kernel(MapperParams mp, const float* __restrict__ params,
                                                             const int* __restrict__ slot, long long slot_stride,
                                                             const float* __restrict__ hmask, float norm_scale,
                                                             const float* __restrict__ word,
                                                             const float* __restrict__ dword_src,
                                                             const int* __restrict__ dword_rows, long long ld_src,
                                                             const float* __restrict__ dbypass,
                                                             const float* __restrict__ save,
                                                             float* __restrict__ rowgrads,
                                                             float* __restrict__ denc) {
  __shared__ __attribute__((aligned(16))) float dout[2048 + 64], part[MT * 4], dz[MAXH], da[MAXH], t2[MAXH];
  __shared__ float red[MT / 64], st[2];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (slot) params += (long long)slot[0] * slot_stride;
  const int E = mp.E, hd = mp.hd, D = mp.D, OD = mp.OD;
  const float* sv = save + (long long)r * (E + 4 * hd + 4);
  float* rg = rowgrads + (long long)r * (OD + 4 * hd);
  // a negative source row = this mapper call reached no prompt position (a prompt without the placeholder: mode-1 captions,
  // negative prompts): its output was never consumed, so every gradient of the row is exactly zero
  const int src_row = dword_rows[r];
  const bool live = src_row >= 0;
  const float* dw = dword_src + (long long)(live ? src_row : 0) * ld_src;
  // ---- through F.normalize * norm_scale ----
  float dot = 0.f;
  if (norm_scale > 0.f)
    for (int o = tid; o < D; o += MT) dot += (word[(long long)r * D + o] / norm_scale) * dw[o];
  dot = block_sum(dot, red);
  const float nrm = fmaxf(sv[E + 4 * hd + 2], 1e-12f);
  for (int o = tid; o < OD; o += MT) {
    float g;
    if (o < D) {
      g = dw[o];
      if (norm_scale > 0.f) g = (norm_scale / nrm) * (g - (word[(long long)r * D + o] / norm_scale) * dot);
    } else {
      g = dbypass ? dbypass[(long long)r * D + (o - D)] : 0.f;
    }
    g = live ? g : 0.f;
    dout[o] = g;
    rg[o] = g;
  }
  __syncthreads();
  // ---- da2m[j] = sum_o Wout[o][j] dout[o] ; LN2 / leaky backward ----
  matvec_cols<VEC>(params + mp.wo, dout, OD, hd, part, t2);
  // mean terms of a LayerNorm backward over the first hd threads' da[] (and da * xhat)
  auto ln_means = [&](const float* xhat) {
    if (tid < 64) {
      float m1 = 0.f, m2 = 0.f;
      for (int i = tid; i < hd; i += 64) {
        m1 += da[i];
        m2 += da[i] * xhat[i];
      }
      m1 = wave_sum(m1);
      m2 = wave_sum(m2);
      if (tid == 0) {
        st[0] = m1 / hd;
        st[1] = m2 / hd;
      }
    }
    __syncthreads();
  };
  if (tid < hd) {
    float t = t2[tid];
    if (hmask) t *= hmask[r * hd + tid];
    // leaky backward needs the pre-activation sign: y2 = xh2*g2 + be2
    const float xh2 = sv[E + 2 * hd + tid];
    const float y2 = xh2 * params[mp.g2 + tid] + params[mp.be2 + tid];
    const float dy = t * (y2 > 0.f ? 1.f : 0.01f);
    rg[OD + hd + tid] = dy;              // dy2
    da[tid] = dy * params[mp.g2 + tid];  // dxhat2
  }
  __syncthreads();
  ln_means(sv + E + 2 * hd);
  if (tid < hd) {
    const float v = sv[E + 4 * hd + 1] * (da[tid] - st[0] - sv[E + 2 * hd + tid] * st[1]);
    dz[tid] = v;
    rg[OD + tid] = v;  // dz2
  }
  __syncthreads();
  // ---- da1[i] = sum_j W3[j][i] dz2[j] ; LN1 / leaky backward ----
  matvec_cols<VEC>(params + mp.w3, dz, hd, hd, part, t2);
  if (tid < hd) {
    const float xh1 = sv[E + tid];
    const float y1 = xh1 * params[mp.g1 + tid] + params[mp.be1 + tid];
    const float dy = t2[tid] * (y1 > 0.f ? 1.f : 0.01f);
    rg[OD + 3 * hd + tid] = dy;  // dy1
    da[tid] = dy * params[mp.g1 + tid];
  }
  __syncthreads();
  ln_means(sv + E);
  if (tid < hd) {
    const float v = sv[E + 4 * hd] * (da[tid] - st[0] - sv[E + tid] * st[1]);
    rg[OD + 2 * hd + tid] = v;  // dz1
    dz[tid] = v;
  }
  if (denc) {  // legacy mapper: gradient w.r.t. the first layer's input, d enc[i] = sum_j W0[j][i] dz1[j]
    __syncthreads();
    matvec_cols<VEC>(params + mp.w0, dz, hd, E, part, t2);
    if (tid < E) denc[(long long)r * E + tid] = t2[tid];
  }
}

// ---------------------------------------------------------------------------------------------
// legacy mapper (arch_view_net <= 14): NeTIPositionalEncoding + the trainable input_layer
//   v = cat[sin(w x), cos(w x)] / |.|  of x = (t, l) RAW (models/positional_encoding.py:23-41); |v| = sqrt(num_w)
//   e = W_in v + b_in,  W_in [E][2*num_w]  (models/neti_mapper.py:155-163, :200-206)
// pin layout: W_in [E][P2] | b_in [E]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float legacy_pe(const float* __restrict__ w_pe, int nw, int k, float t, float l, float inv) {
  const int f = k < nw ? k : k - nw;
  const float p = w_pe[2 * f] * t + w_pe[2 * f + 1] * l;
  return (k < nw ? sinf(p) : cosf(p)) * inv;
}

__global__ __launch_bounds__(256) void legacy_input_fwd_kernel(const float* __restrict__ pin, const int* __restrict__ slot,
                                                               long long slot_stride, const long long* __restrict__ t,
                                                               const float* __restrict__ w_pe, float* __restrict__ enc_out,
                                                               int nl, int Bn, int E, int P2) {
  __shared__ float v[4096];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (slot) pin += (long long)slot[0] * slot_stride;
  const int nw = P2 / 2;
  const float tt = (float)t[r % Bn], ll = (float)(r / Bn), inv = rsqrtf((float)nw);
  for (int k = tid; k < P2; k += 256) v[k] = legacy_pe(w_pe, nw, k, tt, ll, inv);
  __syncthreads();
  for (int o = wave; o < E; o += 4) {  // one wave per output: lanes stride the 2048 inputs (coalesced weight rows)
    const float* wr = pin + (long long)o * P2;
    float s = 0.f;
    for (int k = lane; k < P2; k += 64) s += wr[k] * v[k];
    s = wave_sum(s);
    if (lane == 0) enc_out[(long long)r * E + o] = s + pin[(long long)E * P2 + o];
  }
}

// d W_in[o][k] = sum_r denc[r][o] v_r[k],  d b_in[o] = sum_r denc[r][o]; one block per 256 consecutive k
__global__ __launch_bounds__(256) void legacy_input_bwd_kernel(const long long* __restrict__ t, const float* __restrict__ w_pe,
                                                               const float* __restrict__ denc, float* __restrict__ gin,
                                                               const int* __restrict__ slot, long long slot_stride,
                                                               int accumulate, int nl, int Bn, int E, int P2) {
  extern __shared__ float vs[];  // [R][256]
  const int R = nl * Bn, tid = threadIdx.x, k = blockIdx.x * 256 + tid, nw = P2 / 2;
  if (slot) gin += (long long)slot[0] * slot_stride;
  const float inv = rsqrtf((float)nw);
  if (k < P2)
    for (int r = 0; r < R; ++r) vs[r * 256 + tid] = legacy_pe(w_pe, nw, k, (float)t[r % Bn], (float)(r / Bn), inv);
  __syncthreads();
  if (k < P2) {
    for (int o = 0; o < E; ++o) {
      float s = 0.f;
      for (int r = 0; r < R; ++r) s += denc[(long long)r * E + o] * vs[r * 256 + tid];
      float* g = gin + (long long)o * P2 + k;
      *g = accumulate ? *g + s : s;
    }
  }
  if (blockIdx.x == 0 && tid < E) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += denc[(long long)r * E + tid];
    float* g = gin + (long long)E * P2 + tid;
    *g = accumulate ? *g + s : s;
  }
}


// backward stage 2: one thread per parameter, summing the per-row outer products over R rows.
__global__ __launch_bounds__(256) void mapper_bwd_reduce_kernel(MapperParams mp, int R,
                                                                const float* __restrict__ save,
                                                                const float* __restrict__ rowgrads,
                                                                float* __restrict__ grads, int nparams,
                                                                int accumulate, const int* __restrict__ slot,
                                                                long long slot_stride) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nparams) return;
  if (slot) grads += (long long)slot[0] * slot_stride;
  const int E = mp.E, hd = mp.hd, OD = mp.OD;
  const int ssz = E + 4 * hd + 4, gsz = OD + 4 * hd;
  // which tensor?
  int sa, sb;        // (offset in save row or -1 for "1"), (offset in rowgrads row)
  if (p >= mp.bo) {  // output bias
    sa = -1;
    sb = p - mp.bo;
  } else if (p >= mp.wo) {
    int o = (p - mp.wo) / hd, j = (p - mp.wo) % hd;
    sa = E + 3 * hd + j;  // a2m
    sb = o;
  } else if (p >= mp.be2) {
    sa = -1;
    sb = OD + hd + (p - mp.be2);  // dy2
  } else if (p >= mp.g2) {
    sa = E + 2 * hd + (p - mp.g2);  // xh2
    sb = OD + hd + (p - mp.g2);
  } else if (p >= mp.b3) {
    sa = -1;
    sb = OD + (p - mp.b3);  // dz2
  } else if (p >= mp.w3) {
    int j = (p - mp.w3) / hd, i = (p - mp.w3) % hd;
    sa = E + hd + i;  // a1
    sb = OD + j;
  } else if (p >= mp.be1) {
    sa = -1;
    sb = OD + 3 * hd + (p - mp.be1);  // dy1
  } else if (p >= mp.g1) {
    sa = E + (p - mp.g1);  // xh1
    sb = OD + 3 * hd + (p - mp.g1);
  } else if (p >= mp.b0) {
    sa = -1;
    sb = OD + 2 * hd + (p - mp.b0);  // dz1
  } else {
    int j = (p - mp.w0) / E, i = (p - mp.w0) % E;
    sa = i;  // enc
    sb = OD + 2 * hd + j;
  }
  // eight rows' loads in flight per thread (the loop is a chain of L2 round trips otherwise); same summation order
  float s = 0.f;
  const float* sp = save + (sa >= 0 ? sa : 0);
  const float* gp = rowgrads + sb;
  int r = 0;
  for (; r + 8 <= R; r += 8) {
    float a[8], g[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = sp[(long long)(r + u) * ssz];
      g[u] = gp[(long long)(r + u) * gsz];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (sa >= 0 ? a[u] : 1.f) * g[u];
  }
  for (; r < R; ++r) s += (sa >= 0 ? sp[(long long)r * ssz] : 1.f) * gp[(long long)r * gsz];
  grads[p] = accumulate ? grads[p] + s : s;
}

// ---------------------------------------------------------------------------------------------
// text embeddings: X[(l,b,pos)] = (pos == placeholder pos ? mapper word[(l,b)] : E[id]) + P[pos]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb,
                                                         const long long* __restrict__ ids,
                                                         const int* __restrict__ pos_obj,
                                                         const float* __restrict__ word_obj,
                                                         const int* __restrict__ pos_view,
                                                         const float* __restrict__ word_view, float* __restrict__ X,
                                                         int nl, int Bn, int L, int D) {
  const int d4 = D / 4;
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)nl * Bn * L * d4;
  if (gid >= total) return;
  int c = (int)(gid % d4) * 4;
  long long row = gid / d4;
  int pos = (int)(row % L);
  int b = (int)((row / L) % Bn);
  int l = (int)(row / ((long long)L * Bn));
  const float* src = tok_emb + ids[b * L + pos] * (long long)D;
  if (pos_obj && pos_obj[b] == pos) src = word_obj + ((long long)l * Bn + b) * D;
  if (pos_view && pos_view[b] == pos) src = word_view + ((long long)l * Bn + b) * D;  // view overwrites last, as in the reference
  f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
  f32x4 p = *reinterpret_cast<const f32x4*>(pos_emb + (long long)pos * D + c);
  *reinterpret_cast<f32x4*>(X + row * D + c) = v + p;
}

// ---------------------------------------------------------------------------------------------
// final LayerNorm of the last hidden state (-> key context) and of its bypass-injected clone
// (-> value context).  One wave per row; only placeholder rows differ between the two.
// constrained bypass: new = x + alpha * b/|b| * |x|   (neti_clip_text_encoder.py:138-143)
// ---------------------------------------------------------------------------------------------
constexpr int TMAX = 4;  // chunks of 8 per lane -> D <= 2048

__device__ __forceinline__ void row_load(const float* p, int D, int lane, float v[TMAX][8]) {
#pragma unroll
  for (int i = 0; i < TMAX; ++i) {
    int c = lane + 64 * i;
    if (c * 8 < D) {
      f32x4 a = *reinterpret_cast<const f32x4*>(p + c * 8), b = *reinterpret_cast<const f32x4*>(p + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = a[j];
        v[i][4 + j] = b[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
}
__device__ __forceinline__ float row_dot(const float a[TMAX][8], const float b[TMAX][8]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TMAX; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[i][j] * b[i][j];
  return wave_sum(s);
}
__device__ __forceinline__ void row_stats(const float v[TMAX][8], int D, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TMAX; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[i][j];
  mean = wave_sum(s) / D;
  float q = 0.f;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < TMAX; ++i)
    if ((lane + 64 * i) * 8 < D)
#pragma unroll
      for (int j = 0; j < 8; ++j) q += (v[i][j] - mean) * (v[i][j] - mean);
  rstd = rsqrtf(wave_sum(q) / D + eps);
}
__device__ __forceinline__ void row_ln_store(const float v[TMAX][8], int D, int lane, float mean, float rstd,
                                             const float* gamma, const float* beta, half_t* out) {
#pragma unroll
  for (int i = 0; i < TMAX; ++i) {
    int c = lane + 64 * i;
    if (c * 8 < D) {
      half8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (half_t)((v[i][j] - mean) * rstd * gamma[c * 8 + j] + beta[c * 8 + j]);
      *reinterpret_cast<half8*>(out + c * 8) = h;
    }
  }
}
// dx = LayerNorm input-gradient for upstream gradient dy (f16 row) at input v
__device__ __forceinline__ void row_ln_bwd(const float v[TMAX][8], const half_t* dy, int D, int lane, float eps,
                                           const float* gamma, float g[TMAX][8]) {
  float mean, rstd;
  row_stats(v, D, eps, mean, rstd);
  float xh[TMAX][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < TMAX; ++i) {
    int c = lane + 64 * i;
    if (c * 8 < D) {
      half8 d = *reinterpret_cast<const half8*>(dy + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[i][j] = (v[i][j] - mean) * rstd;
        g[i][j] = (float)d[j] * gamma[c * 8 + j];
        s1 += g[i][j];
        s2 += g[i][j] * xh[i][j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[i][j] = 0.f;
        g[i][j] = 0.f;
      }
    }
  }
  s1 = wave_sum(s1) / D;
  s2 = wave_sum(s2) / D;
#pragma unroll
  for (int i = 0; i < TMAX; ++i)
    if ((lane + 64 * i) * 8 < D)
#pragma unroll
      for (int j = 0; j < 8; ++j) g[i][j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
}

struct BypassArgs {
  const int* pos;      // [B] placeholder position per sample (or null)
  const float* b;      // [nl*B][D] bypass vectors
  float* db;           // [nl*B][D] gradient (backward only)
  float alpha;
  const float* nterm;  // [nl*B] detached mean row norm (unconstrained bypass) or null (constrained)
};

// normalizing terms of the unconstrained bypass (neti_clip_text_encoder.py:145-149,168-172): per
// (layer, sample) the mean over the L token rows of |row|, taken on the tensor *as it is when the
// mapper's turn comes*: the view term sees the object row already replaced.
// nterm[0][(l,b)] = object term, nterm[1][(l,b)] = view term.
__global__ __launch_bounds__(256) void text_norm_terms_kernel(const float* __restrict__ last, BypassArgs obj,
                                                              int obj_unc, float* __restrict__ nterm, int nl, int Bn,
                                                              int L, int D) {
  __shared__ float part[4];
  __shared__ float extra[2];  // |x_p| of the object row, |new object row|
  const int lb = blockIdx.x, b = lb % Bn;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pobj = obj.pos ? obj.pos[b] : -1;
  float s = 0.f;
  for (int pos = wave; pos < L; pos += 4) {
    float x[TMAX][8];
    row_load(last + ((long long)lb * L + pos) * D, D, lane, x);
    const float nx = sqrtf(row_dot(x, x));
    s += nx;
    if (pos == pobj) {
      float nn = 0.f;
      if (!obj_unc) {  // constrained object row: x + alpha * b/|b| * |x|
        float bv[TMAX][8];
        row_load(obj.b + (long long)lb * D, D, lane, bv);
        const float f = obj.alpha * nx / sqrtf(row_dot(bv, bv));
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) bv[i][j] = x[i][j] + f * bv[i][j];
        nn = sqrtf(row_dot(bv, bv));
      }
      if (lane == 0) {
        extra[0] = nx;
        extra[1] = nn;
      }
    }
  }
  if (lane == 0) part[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S = part[0] + part[1] + part[2] + part[3];
    const float mo = S / (float)L;
    nterm[lb] = mo;
    float mv = mo;
    if (pobj >= 0) mv = (S - extra[0] + (obj_unc ? mo : extra[1])) / (float)L;
    nterm[nl * Bn + lb] = mv;
  }
}

__global__ __launch_bounds__(256) void text_final_fwd_kernel(const float* __restrict__ last,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             BypassArgs obj, BypassArgs view,
                                                             half_t* __restrict__ ctx_k, half_t* __restrict__ ctx_v,
                                                             int nl, int Bn, int L, int D) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)nl * Bn * L) return;
  const int pos = (int)(row % L);
  const int b = (int)((row / L) % Bn);
  const int l = (int)(row / ((long long)L * Bn));
  float x[TMAX][8];
  row_load(last + row * D, D, lane, x);
  float mean, rstd;
  row_stats(x, D, eps, mean, rstd);
  row_ln_store(x, D, lane, mean, rstd, gamma, beta, ctx_k + row * D);
  bool changed = false;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const BypassArgs& a = which == 0 ? obj : view;
    if (a.pos && a.pos[b] == pos) {
      float bv[TMAX][8];
      row_load(a.b + ((long long)l * Bn + b) * D, D, lane, bv);
      const float nb = sqrtf(row_dot(bv, bv));
      if (a.nterm) {  // unconstrained: the row is replaced by b/|b| * detach(mean_j |x_j|)
        const float f = a.nterm[l * Bn + b] / nb;
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) x[i][j] = f * bv[i][j];
      } else {
        const float f = a.alpha * sqrtf(row_dot(x, x)) / nb;
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) x[i][j] += f * bv[i][j];
      }
      changed = true;
    }
  }
  if (changed) row_stats(x, D, eps, mean, rstd);
  row_ln_store(x, D, lane, mean, rstd, gamma, beta, ctx_v + row * D);
}

// dX[row] (f32) from dctx_k, dctx_v (f16); placeholder rows also produce d(bypass).
__global__ __launch_bounds__(256) void text_final_bwd_kernel(const float* __restrict__ last,
                                                             const float* __restrict__ gamma, float eps,
                                                             BypassArgs obj, BypassArgs view,
                                                             const half_t* __restrict__ dctx_k,
                                                             const half_t* __restrict__ dctx_v,
                                                             float* __restrict__ dX, int nl, int Bn, int L, int D) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)nl * Bn * L) return;
  const int pos = (int)(row % L);
  const int b = (int)((row / L) % Bn);
  const int l = (int)(row / ((long long)L * Bn));
  float x[TMAX][8], g1[TMAX][8];
  row_load(last + row * D, D, lane, x);
  row_ln_bwd(x, dctx_k + row * D, D, lane, eps, gamma, g1);
  const bool is_obj = obj.pos && obj.pos[b] == pos;
  const bool is_view = view.pos && view.pos[b] == pos;
  float g2[TMAX][8];
  if (!is_obj && !is_view) {
    row_ln_bwd(x, dctx_v + row * D, D, lane, eps, gamma, g2);
  } else {
    // a row is the placeholder of at most one mapper (object and view tokens sit at different positions)
    const BypassArgs& a = is_obj ? obj : view;
    float bv[TMAX][8], nw[TMAX][8];
    row_load(a.b + ((long long)l * Bn + b) * D, D, lane, bv);
    const float nb = sqrtf(row_dot(bv, bv)), nx = sqrtf(row_dot(x, x));
    const bool unc = a.nterm != nullptr;
    const float f = unc ? a.nterm[l * Bn + b] / nb : a.alpha * nx / nb;
#pragma unroll
    for (int i = 0; i < TMAX; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) nw[i][j] = (unc ? 0.f : x[i][j]) + f * bv[i][j];
    float g[TMAX][8];
    row_ln_bwd(nw, dctx_v + row * D, D, lane, eps, gamma, g);
    // constrained:   new = x + alpha * u * |x|, u = b/|b|:  dx = g + alpha (u.g) x/|x| ;  db = alpha |x|/|b| (g - u (u.g))
    // unconstrained: new = m * u with m detached:           dx = 0                     ;  db = m/|b| (g - u (u.g))
    const float ug = row_dot(bv, g) / nb;
    float* dbp = a.db + ((long long)l * Bn + b) * D;
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
      int c = lane + 64 * i;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        g2[i][j] = unc ? 0.f : g[i][j] + a.alpha * ug * x[i][j] / nx;
        if (c * 8 < D) dbp[c * 8 + j] = f * (g[i][j] - bv[i][j] / nb * ug);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TMAX; ++i) {
    int c = lane + 64 * i;
    if (c * 8 < D) {
      f32x4 o0 = {g1[i][0] + g2[i][0], g1[i][1] + g2[i][1], g1[i][2] + g2[i][2], g1[i][3] + g2[i][3]};
      f32x4 o1 = {g1[i][4] + g2[i][4], g1[i][5] + g2[i][5], g1[i][6] + g2[i][6], g1[i][7] + g2[i][7]};
      *reinterpret_cast<f32x4*>(dX + row * D + c * 8) = o0;
      *reinterpret_cast<f32x4*>(dX + row * D + c * 8 + 4) = o1;
    }
  }
}

// data[(l,b)] = [ t_b/1000*2-1, l/nl*2-1, view_params[b][0..nv) ]  (neti_mapper.py:545-562)
__global__ void mapper_inputs_kernel(const long long* __restrict__ t, const float* __restrict__ view_params, int nv,
                                     float* __restrict__ data, int nl, int Bn) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int nf = 2 + nv;
  if (gid >= nl * Bn * nf) return;
  int f = gid % nf;
  int r = gid / nf;
  int b = r % Bn, l = r / Bn;
  float v;
  if (f == 0) v = (float)t[b] / 1000.f * 2.f - 1.f;
  else if (f == 1) v = (float)l / (float)nl * 2.f - 1.f;
  else v = view_params[b * nv + (f - 2)];
  data[gid] = v;
}

__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y,
                                                           long long n8) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n8) return;
  f32x4 a = *reinterpret_cast<const f32x4*>(x + gid * 8), b = *reinterpret_cast<const f32x4*>(x + gid * 8 + 4);
  half8 h = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
  *reinterpret_cast<half8*>(y + gid * 8) = h;
}

int fill_mp(MapperParams& mp, int E, int hd, int D, int has_bypass) {
  if (E <= 0 || E > MAXH || E % 2 || hd <= 0 || hd > MAXH || 256 % hd != 0 || D <= 0) return -1;
  mp.E = E;
  mp.hd = hd;
  mp.D = D;
  mp.OD = has_bypass ? 2 * D : D;
  if (mp.OD > 2048 + 64) return -1;
  int o = 0;
  mp.w0 = o; o += hd * E;
  mp.b0 = o; o += hd;
  mp.g1 = o; o += hd;
  mp.be1 = o; o += hd;
  mp.w3 = o; o += hd * hd;
  mp.b3 = o; o += hd;
  mp.g2 = o; o += hd;
  mp.be2 = o; o += hd;
  mp.wo = o; o += mp.OD * hd;
  mp.bo = o; o += mp.OD;
  return o;
}

// 16-byte weight loads need every matrix of the bucket on a 16-byte boundary (for every slot of a multi-mapper bucket)
// and row lengths that are multiples of 4; MT / (row length / 4) partitions must exist for the transposed products
bool mapper_vec_ok(const MapperParams& mp, const float* params, long long slot_stride) {
  return ((uintptr_t)params & 15) == 0 && slot_stride % 4 == 0 && mp.E % 4 == 0 && mp.hd % 4 == 0 && mp.w0 % 4 == 0 &&
         mp.w3 % 4 == 0 && mp.wo % 4 == 0 && mp.E / 4 <= MT && mp.hd / 4 <= MT;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" long long vneti_mapper_num_params(int enc_dim, int hidden, int D, int has_bypass) {
  MapperParams mp;
  return fill_mp(mp, enc_dim, hidden, D, has_bypass);
}
extern "C" long long vneti_mapper_save_floats(int R, int enc_dim, int hidden) {
  return (long long)R * (enc_dim + 4 * hidden + 4);
}
extern "C" long long vneti_mapper_rowgrad_floats(int R, int hidden, int D, int has_bypass) {
  return (long long)R * ((has_bypass ? 2 * D : D) + 4 * hidden);
}

extern "C" int vneti_mapper_fwd(const float* params, const int* slot, long long slot_stride, const float* data,
                                int nfeat, const float* w_enc,
                                const float* hidden_mask, float norm_scale, float* word, float* bypass, float* save,
                                int R, int enc_dim, int hidden, int D, int has_bypass, const float* enc_in,
                                void* stream) {
  MapperParams mp;
  VN_REQUIRE(fill_mp(mp, enc_dim, hidden, D, has_bypass) > 0, "mapper_fwd: unsupported dims E=%d hd=%d D=%d",
             enc_dim, hidden, D);
  VN_REQUIRE(params && word && save && R > 0 && (enc_in || (data && w_enc && nfeat > 0)) && (!has_bypass || bypass),
             "mapper_fwd: bad arguments");
  if (mapper_vec_ok(mp, params, slot_stride))
    hipLaunchKernelGGL(mapper_fwd_kernel<true>, dim3(R), dim3(MT), 0, ST, mp, params, slot, slot_stride, data, nfeat, w_enc,
                       hidden_mask, norm_scale, word, bypass, save, enc_in);
  else
    hipLaunchKernelGGL(mapper_fwd_kernel<false>, dim3(R), dim3(MT), 0, ST, mp, params, slot, slot_stride, data, nfeat, w_enc,
                       hidden_mask, norm_scale, word, bypass, save, enc_in);
  return vneti_check_launch("mapper_fwd");
}

extern "C" int vneti_mapper_bwd(const float* params, const int* slot, long long slot_stride,
                                const float* hidden_mask, float norm_scale, const float* word,
                                const float* dword_src, const int* dword_rows, long long ld_src,
                                const float* dbypass, const float* save, float* rowgrads, float* grads,
                                int accumulate, int R, int enc_dim, int hidden, int D, int has_bypass,
                                float* denc, void* stream) {
  MapperParams mp;
  int np = fill_mp(mp, enc_dim, hidden, D, has_bypass);
  VN_REQUIRE(np > 0, "mapper_bwd: unsupported dims E=%d hd=%d D=%d", enc_dim, hidden, D);
  VN_REQUIRE(params && word && dword_src && dword_rows && save && rowgrads && grads && R > 0,
             "mapper_bwd: bad arguments");
  if (mapper_vec_ok(mp, params, slot_stride))
    hipLaunchKernelGGL(mapper_bwd_rows_kernel<true>, dim3(R), dim3(MT), 0, ST, mp, params, slot, slot_stride, hidden_mask,
                       norm_scale, word, dword_src, dword_rows, ld_src, dbypass, save, rowgrads, denc);
  else
    hipLaunchKernelGGL(mapper_bwd_rows_kernel<false>, dim3(R), dim3(MT), 0, ST, mp, params, slot, slot_stride, hidden_mask,
                       norm_scale, word, dword_src, dword_rows, ld_src, dbypass, save, rowgrads, denc);
  hipLaunchKernelGGL(mapper_bwd_reduce_kernel, dim3(cdiv(np, 256)), dim3(256), 0, ST, mp, R, save,
                     (const float*)rowgrads, grads, np, accumulate, slot, slot_stride);
  return vneti_check_launch("mapper_bwd");
}

extern "C" long long vneti_mapper_legacy_input_params(int enc_dim, int pe_dim) { return (long long)enc_dim * pe_dim + enc_dim; }

extern "C" int vneti_mapper_legacy_input_fwd(const float* params_in, const int* slot, long long slot_stride,
                                             const void* timesteps_i64, const float* w_pe, float* enc_out, int nl, int Bn,
                                             int enc_dim, int pe_dim, void* stream) {
  VN_REQUIRE(params_in && timesteps_i64 && w_pe && enc_out && nl > 0 && Bn > 0, "mapper_legacy_input_fwd: bad arguments");
  VN_REQUIRE(enc_dim > 0 && enc_dim <= MAXH && pe_dim > 0 && pe_dim <= 4096 && pe_dim % 2 == 0,
             "mapper_legacy_input_fwd: unsupported dims E=%d P=%d", enc_dim, pe_dim);
  hipLaunchKernelGGL(legacy_input_fwd_kernel, dim3(nl * Bn), dim3(256), 0, ST, params_in, slot, slot_stride,
                     (const long long*)timesteps_i64, w_pe, enc_out, nl, Bn, enc_dim, pe_dim);
  return vneti_check_launch("mapper_legacy_input_fwd");
}

extern "C" int vneti_mapper_legacy_input_bwd(const void* timesteps_i64, const float* w_pe, const float* denc, float* grads_in,
                                             const int* slot, long long slot_stride, int accumulate, int nl, int Bn,
                                             int enc_dim, int pe_dim, void* stream) {
  VN_REQUIRE(timesteps_i64 && w_pe && denc && grads_in && nl > 0 && Bn > 0, "mapper_legacy_input_bwd: bad arguments");
  VN_REQUIRE(enc_dim > 0 && enc_dim <= 256 && pe_dim > 0 && pe_dim % 2 == 0 && nl * Bn <= 128,
             "mapper_legacy_input_bwd: unsupported dims E=%d P=%d R=%d", enc_dim, pe_dim, nl * Bn);
  const size_t lds = (size_t)nl * Bn * 256 * sizeof(float);
  // > 64 KiB of dynamic LDS needs the function attribute — per DEVICE (one process may drive several), and only when
  // the launch really needs it (nl * Bn > 64 rows); checked, so a part with 64 KiB per workgroup fails here, loudly
  if (lds > 64 * 1024) {
    static bool lds_opt_in[64] = {};
    int dev = 0;
    VN_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "mapper_legacy_input_bwd: hipGetDevice failed");
    if (!lds_opt_in[dev]) {
      hipError_t e = hipFuncSetAttribute((const void*)legacy_input_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         128 * 256 * 4);
      VN_REQUIRE(e == hipSuccess, "mapper_legacy_input_bwd: %d rows need %zu bytes of LDS: %s", nl * Bn, lds,
                 hipGetErrorString(e));
      lds_opt_in[dev] = true;
    }
  }
  hipLaunchKernelGGL(legacy_input_bwd_kernel, dim3(cdiv(pe_dim, 256)), dim3(256), lds, ST, (const long long*)timesteps_i64,
                     w_pe, denc, grads_in, slot, slot_stride, accumulate, nl, Bn, enc_dim, pe_dim);
  return vneti_check_launch("mapper_legacy_input_bwd");
}


extern "C" int vneti_text_embed(const float* tok_emb, const float* pos_emb, const void* ids, const int* pos_obj,
                                const float* word_obj, const int* pos_view, const float* word_view, float* X, int nl,
                                int Bn, int L, int D, void* stream) {
  VN_REQUIRE(tok_emb && pos_emb && ids && X && nl > 0 && Bn > 0 && L > 0 && D % 4 == 0, "text_embed: bad arguments");
  VN_REQUIRE(!(pos_obj && !word_obj) && !(pos_view && !word_view), "text_embed: placeholder positions without words");
  long long n = (long long)nl * Bn * L * (D / 4);
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, tok_emb, pos_emb,
                     (const long long*)ids, pos_obj, word_obj, pos_view, word_view, X, nl, Bn, L, D);
  return vneti_check_launch("text_embed");
}

extern "C" int vneti_text_final_fwd(const float* last, const float* gamma, const float* beta, float eps,
                                    const int* pos_obj, const float* bypass_obj, float alpha_obj,
                                    int unconstrained_obj, const int* pos_view, const float* bypass_view,
                                    float alpha_view, int unconstrained_view, float* norm_terms, void* ctx_k,
                                    void* ctx_v, int nl, int Bn, int L, int D, void* stream) {
  VN_REQUIRE(last && gamma && beta && ctx_k && ctx_v && D % 8 == 0 && D <= 8 * 64 * TMAX, "text_final_fwd: bad arguments");
  const bool uo = unconstrained_obj && pos_obj && bypass_obj, uv = unconstrained_view && pos_view && bypass_view;
  VN_REQUIRE(!(uo || uv) || norm_terms, "text_final_fwd: unconstrained bypass needs norm_terms[2*nl*B]");
  BypassArgs o{pos_obj && bypass_obj ? pos_obj : nullptr, bypass_obj, nullptr, alpha_obj, uo ? norm_terms : nullptr};
  BypassArgs v{pos_view && bypass_view ? pos_view : nullptr, bypass_view, nullptr, alpha_view,
               uv ? norm_terms + (long long)nl * Bn : nullptr};
  if (uo || uv)
    hipLaunchKernelGGL(text_norm_terms_kernel, dim3(nl * Bn), dim3(256), 0, ST, last, o, uo ? 1 : 0, norm_terms, nl,
                       Bn, L, D);
  long long rows = (long long)nl * Bn * L;
  hipLaunchKernelGGL(text_final_fwd_kernel, dim3((unsigned)cdivl(rows, 4)), dim3(256), 0, ST, last, gamma, beta, eps, o,
                     v, (half_t*)ctx_k, (half_t*)ctx_v, nl, Bn, L, D);
  return vneti_check_launch("text_final_fwd");
}

extern "C" int vneti_text_final_bwd(const float* last, const float* gamma, float eps, const int* pos_obj,
                                    const float* bypass_obj, float alpha_obj, int unconstrained_obj,
                                    float* dbypass_obj, const int* pos_view, const float* bypass_view,
                                    float alpha_view, int unconstrained_view, float* dbypass_view,
                                    const float* norm_terms, const void* dctx_k, const void* dctx_v, float* dX,
                                    int nl, int Bn, int L, int D, void* stream) {
  VN_REQUIRE(last && gamma && dctx_k && dctx_v && dX && D % 8 == 0 && D <= 8 * 64 * TMAX, "text_final_bwd: bad arguments");
  VN_REQUIRE(!(pos_obj && bypass_obj && !dbypass_obj) && !(pos_view && bypass_view && !dbypass_view),
             "text_final_bwd: missing bypass gradient buffer");
  const bool uo = unconstrained_obj && pos_obj && bypass_obj, uv = unconstrained_view && pos_view && bypass_view;
  VN_REQUIRE(!(uo || uv) || norm_terms, "text_final_bwd: unconstrained bypass needs the forward's norm_terms");
  BypassArgs o{pos_obj && bypass_obj ? pos_obj : nullptr, bypass_obj, dbypass_obj, alpha_obj, uo ? norm_terms : nullptr};
  BypassArgs v{pos_view && bypass_view ? pos_view : nullptr, bypass_view, dbypass_view, alpha_view,
               uv ? norm_terms + (long long)nl * Bn : nullptr};
  long long rows = (long long)nl * Bn * L;
  hipLaunchKernelGGL(text_final_bwd_kernel, dim3((unsigned)cdivl(rows, 4)), dim3(256), 0, ST, last, gamma, eps, o, v,
                     (const half_t*)dctx_k, (const half_t*)dctx_v, dX, nl, Bn, L, D);
  return vneti_check_launch("text_final_bwd");
}

extern "C" int vneti_mapper_inputs(const void* timesteps, const float* view_params, int nv, float* data, int nl,
                                   int Bn, void* stream) {
  VN_REQUIRE(timesteps && data && nl > 0 && Bn > 0 && nv >= 0 && (nv == 0 || view_params), "mapper_inputs: bad arguments");
  int n = nl * Bn * (2 + nv);
  hipLaunchKernelGGL(mapper_inputs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const long long*)timesteps,
                     view_params, nv, data, nl, Bn);
  return vneti_check_launch("mapper_inputs");
}

extern "C" int vneti_cast_f32_f16(const float* x, void* y, long long n, void* stream) {
  VN_REQUIRE(x && y && n > 0 && n % 8 == 0, "cast_f32_f16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)cdivl(n / 8, 256)), dim3(256), 0, ST, x, (half_t*)y, n / 8);
  return vneti_check_launch("cast_f32_f16");
}
