// Small HBM/latency-bound kernels around the GEMMs: residual add, GEGLU, quick-GELU, sinusoidal
// timestep embedding, 2x2 gradient pooling (nearest-upsample backward), the fused latent
// sampling + DDPM add-noise, MSE loss + its gradient seed, AdamW with GradScaler semantics and
// the device-side RNG that keeps the whole train step hipGraph-capturable.
//
// Reference call sites: training/coach.py:165-183 (latent sample, randn_like, randint,
// add_noise), :201-214 (target, mse_loss, backward), :216-218 (AdamW step) and diffusers'
// GEGLU / Timesteps / Upsample2D inside `self.unet(...)` (:197).
#include "common.h"
#include "../../include/vneti.h"

namespace {

__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// counter-based standard normal (Box-Muller on two hashed uniforms)
__device__ __forceinline__ float rng_normal(uint32_t seed, uint32_t ctr, uint32_t idx) {
  uint32_t a = hash_u32(idx * 0x9E3779B1U + seed);
  a = hash_u32(a ^ (ctr * 0x85EBCA6BU + 0x632BE5ABU));
  uint32_t b = hash_u32(a + 0x68E31DA4U);
  float u1 = ((a >> 8) + 1u) * (1.0f / 16777216.0f);  // (0, 1]
  float u2 = (b >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(256) void add_kernel(const half_t* a, long long lda, const half_t* b, long long ldb,
                                                  half_t* out, long long ldo, int rows, int cchunks) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * cchunks) return;
  int r = (int)(gid / cchunks), c = (int)(gid - (long long)r * cchunks) * 8;
  half8 x = *reinterpret_cast<const half8*>(a + (long long)r * lda + c);
  half8 y = *reinterpret_cast<const half8*>(b + (long long)r * ldb + c);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)x[j] + (float)y[j]);
  vn_st16_wt(vn_make_rsrc(out, 0x7fffffffu), (uint32_t)(((long long)r * ldo + c) * 2), o);
}

// GEGLU: p = [h | g] (each C4 wide); out = h * gelu_erf(g)
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const half_t* p, long long ldp, half_t* out, long long ldo,
                                                        int rows, int C4) {
  int cch = C4 / 8;
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * cch) return;
  int r = (int)(gid / cch), c = (int)(gid - (long long)r * cch) * 8;
  half8 h = *reinterpret_cast<const half8*>(p + (long long)r * ldp + c);
  half8 g = *reinterpret_cast<const half8*>(p + (long long)r * ldp + C4 + c);
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)h[j] * vn_gelu_erf((float)g[j]));
  *reinterpret_cast<half8*>(out + (long long)r * ldo + c) = o;
}
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const half_t* dy, long long lddy, const half_t* p,
                                                        long long ldp, half_t* dp, long long lddp, int rows,
                                                        int C4) {
  int cch = C4 / 8;
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)rows * cch) return;
  int r = (int)(gid / cch), c = (int)(gid - (long long)r * cch) * 8;
  half8 d = *reinterpret_cast<const half8*>(dy + (long long)r * lddy + c);
  half8 h = *reinterpret_cast<const half8*>(p + (long long)r * ldp + c);
  half8 g = *reinterpret_cast<const half8*>(p + (long long)r * ldp + C4 + c);
  half8 dh, dg;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float gv = (float)g[j], dv = (float)d[j], hv = (float)h[j];
    float cdf, xpdf;
    vn_gelu_parts(gv, cdf, xpdf);
    dh[j] = (half_t)(dv * gv * cdf);
    dg[j] = (half_t)(dv * hv * (cdf + xpdf));
  }
  *reinterpret_cast<half8*>(dp + (long long)r * lddp + c) = dh;
  *reinterpret_cast<half8*>(dp + (long long)r * lddp + C4 + c) = dg;
}

// y = act(x) and dx = dy * act'(x) for the CLIP MLP (quick-GELU or exact GELU), f16
__global__ __launch_bounds__(256) void act_fwd_kernel(const half_t* x, half_t* y, long long n8, int act) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n8) return;
  half8 v = *reinterpret_cast<const half8*>(x + gid * 8), o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float f = (float)v[j];
    o[j] = (half_t)(act == 2 ? vn_quick_gelu(f) : (act == 3 ? vn_gelu_erf(f) : vn_silu(f)));
  }
  *reinterpret_cast<half8*>(y + gid * 8) = o;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const half_t* dy, const half_t* x, half_t* dx, long long n8,
                                                      int act) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n8) return;
  half8 d = *reinterpret_cast<const half8*>(dy + gid * 8);
  half8 v = *reinterpret_cast<const half8*>(x + gid * 8), o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float f = (float)v[j], g;
    if (act == 2) {
      float s = vn_sigmoid(1.702f * f);
      g = s * (1.f + 1.702f * f * (1.f - s));
    } else if (act == 3) {
      g = vn_gelu_erf_grad(f);
    } else {
      float s = vn_sigmoid(f);
      g = s * (1.f + f * (1.f - s));
    }
    o[j] = (half_t)((float)d[j] * g);
  }
  *reinterpret_cast<half8*>(dx + gid * 8) = o;
}

// Timesteps(dim, flip_sin_to_cos=True, freq_shift=0): out[b] = [cos(t f_i) | sin(t f_i)], f16
__global__ void timestep_embedding_kernel(const long long* t, half_t* out, int Bn, int dim) {
  int half_dim = dim / 2;
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= Bn * half_dim) return;
  int b = gid / half_dim, i = gid - b * half_dim;
  float f = __expf(-9.210340371976184f * (float)i / (float)half_dim);
  float e = (float)t[b] * f;
  out[(long long)b * dim + i] = (half_t)cosf(e);
  out[(long long)b * dim + half_dim + i] = (half_t)sinf(e);
}

// out[b][y][x][:] = sum of the 2x2 block of in[b][2y..][2x..][:]  (nearest-2x upsample backward)
__global__ __launch_bounds__(256) void sum2x2_kernel(const half_t* in, long long ldi, half_t* out, long long ldo,
                                                     int Bn, int H, int Wd, int cch) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  long long total = (long long)Bn * H * Wd * cch;
  if (gid >= total) return;
  int c = (int)(gid % cch) * 8;
  long long pix = gid / cch;
  int x = (int)(pix % Wd);
  long long t2 = pix / Wd;
  int y = (int)(t2 % H), b = (int)(t2 / H);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      long long ip = ((long long)b * 2 * H + 2 * y + dy) * (2 * Wd) + 2 * x + dx;
      half8 v = *reinterpret_cast<const half8*>(in + ip * ldi + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
    }
  half8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
  vn_st16_wt(vn_make_rsrc(out, 0x7fffffffu), (uint32_t)((pix * ldo + c) * 2), o);
}

// ---- device RNG -----------------------------------------------------------------------------
// state[0] = seed, state[1] = step counter (advanced by rng_advance_kernel once per step)
__global__ void fill_normal_kernel(float* out, long long n, const uint32_t* state, uint32_t stream_id) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  out[gid] = rng_normal(state[0] + stream_id * 0x9E3779B9U, state[1], (uint32_t)gid);
}
__global__ void fill_randint_kernel(long long* out, int n, int high, const uint32_t* state, uint32_t stream_id) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  uint32_t h = hash_u32((uint32_t)gid * 0x9E3779B1U + state[0] + stream_id * 0x9E3779B9U);
  h = hash_u32(h ^ (state[1] * 0x85EBCA6BU + 0x27D4EB2FU));
  out[gid] = (long long)(h % (uint32_t)high);
}
// nested dropout (models/neti_mapper.py:401-414): one Bernoulli(prob) draw per mapper call (= per UNet
// layer l); when it fires every sample b gets its own truncation index ~ U{0..hidden-1} and the
// hidden vector is zeroed from there on.  mask[(l,b)][j] = 1 if kept.
__global__ void nested_dropout_mask_kernel(float* mask, int nl, int Bn, int hidden, float prob,
                                           const uint32_t* state, uint32_t stream_id) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nl * Bn * hidden) return;
  const int j = gid % hidden, r = gid / hidden, l = r / Bn;
  const uint32_t seed = state[0] + stream_id * 0x9E3779B9U;
  uint32_t hl = hash_u32(hash_u32((uint32_t)l * 0x9E3779B1U + seed) ^ (state[1] * 0x85EBCA6BU + 0x27D4EB2FU));
  const bool fire = (float)(hl >> 8) * (1.f / 16777216.f) < prob;
  uint32_t hr = hash_u32(hash_u32((uint32_t)(r + 0x10000) * 0x9E3779B1U + seed) ^ (state[1] * 0x85EBCA6BU + 0x27D4EB2FU));
  const int idx = (int)(hr % (uint32_t)hidden);
  mask[gid] = (!fire || j < idx) ? 1.f : 0.f;
}
__global__ void rng_advance_kernel(uint32_t* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[1] += 1u;
}

// ---- latent sampling + add-noise ----------------------------------------------------------------
// moments: NHWC f16 [B][h][w][2*Lc] (mean | logvar).  eps, noise: f32 [B][Lc][h][w] (NCHW, the
// layout torch.randn_like(latents) has in the reference).  Outputs (all NCHW f32):
//   latents = (mean + exp(0.5*clamp(logvar,-30,20))*eps) * scaling
//   noisy   = sqrt(ac[t]) * latents + sqrt(1-ac[t]) * noise
//   target  = noise (epsilon) or sqrt(ac)*noise - sqrt(1-ac)*latents (v_prediction)
// (the three steps are device helpers so that the split entry points below — vneti_latent_sample / vneti_add_noise, the
// module-call seam of compat/sd_modules.py — round exactly as the fused kernel does)
__device__ __forceinline__ float vn_latent_sample(const half_t* m, int c, int Lc, float eps, float scaling) {
  float mean = (float)m[c];
  float logvar = fminf(fmaxf((float)m[Lc + c], -30.f), 20.f);
  return fmaf(__expf(0.5f * logvar), eps, mean) * scaling;  // explicit fma: the same rounding in every kernel that inlines this
}
__device__ __forceinline__ void vn_add_noise(float z, float n, float a, int vpred, float& noisy, float& target) {
  float sa = sqrtf(a), sb = sqrtf(1.f - a);
  noisy = fmaf(sa, z, sb * n);
  target = vpred ? fmaf(sa, n, -(sb * z)) : n;
}

__global__ void sample_add_noise_kernel(const half_t* moments, long long ldm, const float* eps,
                                        const float* noise, const long long* t, const float* ac, float scaling,
                                        int vpred, float* latents, float* noisy, float* target, int Bn, int Lc,
                                        int HW) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int total = Bn * Lc * HW;
  if (gid >= total) return;
  int p = gid % HW;
  int c = (gid / HW) % Lc;
  int b = gid / (HW * Lc);
  float z = vn_latent_sample(moments + ((long long)b * HW + p) * ldm, c, Lc, eps[gid], scaling);
  float ny, tg;
  vn_add_noise(z, noise[gid], ac[t[b]], vpred, ny, tg);
  latents[gid] = z;
  noisy[gid] = ny;
  target[gid] = tg;
}

// latent_dist.sample() (* scaling) alone: AutoencoderKL.encode(x).latent_dist.sample() of the module-call seam
__global__ void latent_sample_kernel(const half_t* moments, long long ldm, const float* eps, float scaling,
                                     float* latents, int Bn, int Lc, int HW) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= Bn * Lc * HW) return;
  int p = gid % HW;
  int c = (gid / HW) % Lc;
  int b = gid / (HW * Lc);
  latents[gid] = vn_latent_sample(moments + ((long long)b * HW + p) * ldm, c, Lc, eps[gid], scaling);
}

// DDPMScheduler.add_noise / get_velocity alone (noisy and / or target may be null)
__global__ void add_noise_kernel(const float* latents, const float* noise, const long long* t, const float* ac, int vpred,
                                 float* noisy, float* target, int Bn, int Lc, int HW) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= Bn * Lc * HW) return;
  int b = gid / (HW * Lc);
  float ny, tg;
  vn_add_noise(latents[gid], noise[gid], ac[t[b]], vpred, ny, tg);
  if (noisy) noisy[gid] = ny;
  if (target) target[gid] = tg;
}

// ---- inference: classifier-free guidance + one sampler step (sd_pipeline_call.py:72-103) ---------------
// pred: NHWC f16 [2*B*HW][ldp], rows [0,B*HW) unconditional, [B*HW,2*B*HW) conditional.  x, m_prev: NCHW f32
// [B][Lc][HW].  e = u + g (c-u);  x0 = (x - sigma_t e)/alpha_t (epsilon) or alpha_t x - sigma_t e (v);
// x <- cx x + c0 x0 + c1 m_prev;  m_prev <- x0;  x_in (NCHW f32 [2B][Lc][HW], the UNet input) <- x in both halves.
__global__ void cfg_sampler_step_kernel(const half_t* pred, long long ldp, float* x, float* m_prev, float* x_in,
                                        int Bn, int Lc, int HW, float guidance, float alpha_t, float sigma_t, float cx,
                                        float c0, float c1, int vpred) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int total = Bn * Lc * HW;
  if (gid >= total) return;
  int p = gid % HW;
  int c = (gid / HW) % Lc;
  int b = gid / (HW * Lc);
  float u = (float)pred[((long long)b * HW + p) * ldp + c];
  float cnd = (float)pred[((long long)(Bn + b) * HW + p) * ldp + c];
  float e = u + guidance * (cnd - u);
  float xv = x[gid];
  float x0 = vpred ? (alpha_t * xv - sigma_t * e) : (xv - sigma_t * e) / alpha_t;
  float xn = cx * xv + c0 * x0 + c1 * m_prev[gid];
  x[gid] = xn;
  m_prev[gid] = x0;
  x_in[gid] = xn;
  x_in[(long long)total + gid] = xn;
}
// graph-replayable variant: the per-step scalars {alpha_t, sigma_t, cx, c0, c1} come from a device table row
// selected by a device step counter, so ONE captured sampler step replays for every timestep
__global__ void cfg_sampler_step_table_kernel(const half_t* pred, long long ldp, float* x, float* m_prev, float* x_in,
                                              int Bn, int Lc, int HW, float guidance, const float* coef_table,
                                              const int* step, int vpred) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int total = Bn * Lc * HW;
  if (gid >= total) return;
  const float* cf = coef_table + 5 * step[0];
  const float alpha_t = cf[0], sigma_t = cf[1], cx = cf[2], c0 = cf[3], c1 = cf[4];
  int p = gid % HW;
  int c = (gid / HW) % Lc;
  int b = gid / (HW * Lc);
  float u = (float)pred[((long long)b * HW + p) * ldp + c];
  float cnd = (float)pred[((long long)(Bn + b) * HW + p) * ldp + c];
  float e = u + guidance * (cnd - u);
  float xv = x[gid];
  float x0 = vpred ? (alpha_t * xv - sigma_t * e) : (xv - sigma_t * e) / alpha_t;
  float xn = cx * xv + c0 * x0 + c1 * m_prev[gid];
  x[gid] = xn;
  m_prev[gid] = x0;
  x_in[gid] = xn;
  x_in[(long long)total + gid] = xn;
}
__global__ void table_fill_i64_kernel(long long* dst, int n, const long long* table, const int* step) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) dst[gid] = table[step[0]];
}
__global__ void counter_advance_kernel(int* c) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1;
}
// tiny 1x1 conv on NCHW f32 latents (AutoencoderKL.post_quant_conv after the 1/scaling_factor of
// pipeline.decode_latents): out[b][o][p] = bias[o] + sum_c W[o][c] * x[b][c][p] * in_scale,  C <= 8
__global__ void conv1x1_nchw_kernel(const float* x, const float* W, const float* bias, float* out, int Bn, int Ci,
                                    int Co, int HW, float in_scale) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= Bn * Co * HW) return;
  int p = gid % HW;
  int o = (gid / HW) % Co;
  int b = gid / (HW * Co);
  float acc = bias ? bias[o] : 0.f;
  for (int c = 0; c < Ci; ++c) acc += W[o * Ci + c] * x[((long long)b * Ci + c) * HW + p] * in_scale;
  out[gid] = acc;
}
// decoder output NHWC f16 [B*HW][ldi] (3 channels) -> (v/2 + 0.5).clamp(0,1) as f32 [B][HW][3]
// (pipeline.decode_latents, sd_pipeline_call.py:115)
__global__ void image_postprocess_kernel(const half_t* img, long long ldi, float* out, long long n_pix, int ch) {
  long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n_pix * ch) return;
  long long p = gid / ch;
  int c = (int)(gid - p * ch);
  float v = (float)img[p * ldi + c] * 0.5f + 0.5f;
  out[gid] = fminf(fmaxf(v, 0.f), 1.f);
}

// ---- MSE loss + gradient seed -------------------------------------------------------------------
// pred: NHWC f16 [B*HW][ldp] (Lc channels used); target NCHW f32.  loss_sum += sum (p-t)^2 (one
// atomic per block); dpred (NHWC f16) = 2*(p-t)/N * loss_scale[0].
__global__ __launch_bounds__(256) void mse_loss_grad_kernel(const half_t* pred, long long ldp, const float* target,
                                                            half_t* dpred, long long lddp, float* loss_sum,
                                                            const float* loss_scale, int Bn, int Lc, int HW) {
  __shared__ float red[4];
  int gid = blockIdx.x * 256 + threadIdx.x;
  int total = Bn * Lc * HW;
  float sq = 0.f;
  if (gid < total) {
    int c = gid % Lc;
    int p = (gid / Lc) % HW;
    int b = gid / (Lc * HW);
    float pv = (float)pred[((long long)b * HW + p) * ldp + c];
    float tv = target[((long long)b * Lc + c) * HW + p];
    float d = pv - tv;
    sq = d * d;
    dpred[((long long)b * HW + p) * lddp + c] = (half_t)(2.f * d / (float)total * loss_scale[0]);
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

// ---- AdamW over a flat f32 bucket with torch.cuda.amp.GradScaler semantics ---------------------
// scaler[0] = loss scale, scaler[1] = growth tracker, scaler[2] = found_inf (this step),
// hyper: lr, beta1, beta2, eps, weight_decay, grad_div (= world size for DP mean)
__global__ __launch_bounds__(256) void grads_check_finite_kernel(const float* g, long long n, float* scaler) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (gid < n) {
    float v = g[gid];
    bad = !(v == v) || fabsf(v) == INFINITY;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) scaler[2] = 1.f;
}
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, long long n,
                                                    const float* hyper, const float* scaler, const int* step) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  if (scaler[2] != 0.f) return;  // GradScaler.step(): skip the update when grads are non-finite
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gdiv = hyper[5];
  const int t = step[0] + 1;
  float grad = g[gid] / (scaler[0] * gdiv);
  float pv = p[gid] * (1.f - lr * wd);
  float mv = b1 * m[gid] + (1.f - b1) * grad;
  float vv = b2 * v[gid] + (1.f - b2) * grad * grad;
  float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(b2, (float)t);
  float denom = sqrtf(vv) / sqrtf(bc2) + eps;
  p[gid] = pv - (lr / bc1) * mv / denom;
  m[gid] = mv;
  v[gid] = vv;
}
// AdamW over a bucket of equal-length segments (one per mapper) with torch's per-parameter state:
// a segment joins the update set the first time it receives a gradient (grad None -> skipped,
// torch/optim/adamw.py) and from then on is stepped every iteration, with a zero gradient when it
// was not in the batch (zero_grad() keeps zero tensors in torch 1.13), each with its own `step`.
__global__ __launch_bounds__(256) void adamw_segments_kernel(float* p, const float* g, float* m, float* v,
                                                             long long n, long long seg_len, const int* seg_step,
                                                             const int* active, int n_active, const float* hyper,
                                                             const float* scaler) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n) return;
  if (scaler[2] != 0.f) return;
  const int seg = (int)(gid / seg_len);
  bool is_active = false;
  for (int i = 0; i < n_active; ++i) is_active |= active[i] == seg;
  const int t0 = seg_step[seg];
  if (t0 == 0 && !is_active) return;  // never had a gradient: not in the optimizer's update set yet
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], gdiv = hyper[5];
  const int t = t0 + 1;
  float grad = is_active ? g[gid] / (scaler[0] * gdiv) : 0.f;
  float pv = p[gid] * (1.f - lr * wd);
  float mv = b1 * m[gid] + (1.f - b1) * grad;
  float vv = b2 * v[gid] + (1.f - b2) * grad * grad;
  float bc1 = 1.f - powf(b1, (float)t), bc2 = 1.f - powf(b2, (float)t);
  float denom = sqrtf(vv) / sqrtf(bc2) + eps;
  p[gid] = pv - (lr / bc1) * mv / denom;
  m[gid] = mv;
  v[gid] = vv;
}
__global__ void segments_step_kernel(int* seg_step, int n_seg, const int* active, int n_active, const float* scaler) {
  int seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= n_seg || scaler[2] != 0.f) return;
  bool is_active = false;
  for (int i = 0; i < n_active; ++i) is_active |= active[i] == seg;
  if (seg_step[seg] > 0 || is_active) seg_step[seg] += 1;
}
// finite check restricted to the active segments (the others hold stale or zero gradients)
__global__ __launch_bounds__(256) void grads_check_finite_segments_kernel(const float* g, long long seg_len,
                                                                          const int* active, float* scaler) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (i < seg_len) {
    float x = g[(long long)active[blockIdx.y] * seg_len + i];
    bad = !(fabsf(x) <= 3.0e38f);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) scaler[2] = 1.f;
}
// GradScaler.update(): backoff 0.5 on inf, growth x2 every `interval` clean steps; advance step.
// growth_interval <= 0: a STATIC scale (the bf16 branch: accelerate creates no GradScaler there) — a non-finite step is
// still skipped, but the scale neither backs off nor grows
__global__ void scaler_update_kernel(float* scaler, int* step, int growth_interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (growth_interval <= 0) {
    if (scaler[2] == 0.f) step[0] += 1;
  } else if (scaler[2] != 0.f) {
    scaler[0] *= 0.5f;
    scaler[1] = 0.f;
  } else {
    step[0] += 1;
    scaler[1] += 1.f;
    if ((int)scaler[1] >= growth_interval) {
      scaler[0] *= 2.f;
      scaler[1] = 0.f;
    }
  }
  scaler[2] = 0.f;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int vneti_add_f16(const void* a, long long lda, const void* b, long long ldb, void* out, long long ldo,
                             int rows, int cols, void* stream) {
  VN_REQUIRE(a && b && out && rows > 0 && cols > 0 && cols % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0,
             "add_f16: bad arguments");
  long long n = (long long)rows * (cols / 8);
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, (const half_t*)a, lda,
                     (const half_t*)b, ldb, (half_t*)out, ldo, rows, cols / 8);
  return vneti_check_launch("add_f16");
}

extern "C" int vneti_geglu_fwd(const void* p, long long ldp, void* out, long long ldo, int rows, int C4,
                               void* stream) {
  VN_REQUIRE(p && out && rows > 0 && C4 > 0 && C4 % 8 == 0 && ldp % 8 == 0 && ldo % 8 == 0, "geglu_fwd: bad arguments");
  long long n = (long long)rows * (C4 / 8);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, (const half_t*)p, ldp,
                     (half_t*)out, ldo, rows, C4);
  return vneti_check_launch("geglu_fwd");
}

extern "C" int vneti_geglu_bwd(const void* dy, long long lddy, const void* p, long long ldp, void* dp,
                               long long lddp, int rows, int C4, void* stream) {
  VN_REQUIRE(dy && p && dp && rows > 0 && C4 > 0 && C4 % 8 == 0 && ldp % 8 == 0 && lddy % 8 == 0 && lddp % 8 == 0,
             "geglu_bwd: bad arguments");
  long long n = (long long)rows * (C4 / 8);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, (const half_t*)dy, lddy,
                     (const half_t*)p, ldp, (half_t*)dp, lddp, rows, C4);
  return vneti_check_launch("geglu_bwd");
}

extern "C" int vneti_act_fwd_f16(const void* x, void* y, long long n, int act, void* stream) {
  VN_REQUIRE(x && y && n > 0 && n % 8 == 0 && act >= 1 && act <= 3, "act_fwd: bad arguments");
  hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)cdivl(n / 8, 256)), dim3(256), 0, ST, (const half_t*)x,
                     (half_t*)y, n / 8, act);
  return vneti_check_launch("act_fwd");
}

extern "C" int vneti_act_bwd_f16(const void* dy, const void* x, void* dx, long long n, int act, void* stream) {
  VN_REQUIRE(dy && x && dx && n > 0 && n % 8 == 0 && act >= 1 && act <= 3, "act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)cdivl(n / 8, 256)), dim3(256), 0, ST, (const half_t*)dy,
                     (const half_t*)x, (half_t*)dx, n / 8, act);
  return vneti_check_launch("act_bwd");
}

extern "C" int vneti_timestep_embedding(const void* t, void* out, int Bn, int dim, void* stream) {
  VN_REQUIRE(t && out && Bn > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments");
  int n = Bn * dim / 2;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const long long*)t,
                     (half_t*)out, Bn, dim);
  return vneti_check_launch("timestep_embedding");
}

extern "C" int vneti_sum2x2_f16(const void* in, long long ldi, void* out, long long ldo, int Bn, int H, int W,
                                int C, void* stream) {
  VN_REQUIRE(in && out && Bn > 0 && H > 0 && W > 0 && C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0,
             "sum2x2: bad arguments");
  long long n = (long long)Bn * H * W * (C / 8);
  hipLaunchKernelGGL(sum2x2_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, (const half_t*)in, ldi,
                     (half_t*)out, ldo, Bn, H, W, C / 8);
  return vneti_check_launch("sum2x2");
}

extern "C" int vneti_rng_fill_normal(void* out, long long n, const void* state, unsigned stream_id, void* stream) {
  VN_REQUIRE(out && state && n > 0 && n < 0xffffffffLL, "rng_fill_normal: bad arguments");
  hipLaunchKernelGGL(fill_normal_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, (float*)out, n,
                     (const uint32_t*)state, stream_id);
  return vneti_check_launch("rng_fill_normal");
}

extern "C" int vneti_rng_fill_randint(void* out, int n, int high, const void* state, unsigned stream_id,
                                      void* stream) {
  VN_REQUIRE(out && state && n > 0 && high > 0, "rng_fill_randint: bad arguments");
  hipLaunchKernelGGL(fill_randint_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (long long*)out, n, high,
                     (const uint32_t*)state, stream_id);
  return vneti_check_launch("rng_fill_randint");
}

extern "C" int vneti_nested_dropout_mask(float* mask, int nl, int Bn, int hidden, float prob, const void* state,
                                         unsigned stream_id, void* stream) {
  VN_REQUIRE(mask && state && nl > 0 && Bn > 0 && hidden > 0 && prob >= 0.f && prob <= 1.f,
             "nested_dropout_mask: bad arguments");
  int n = nl * Bn * hidden;
  hipLaunchKernelGGL(nested_dropout_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, mask, nl, Bn, hidden, prob,
                     (const uint32_t*)state, (uint32_t)stream_id);
  return vneti_check_launch("nested_dropout_mask");
}

extern "C" int vneti_rng_advance(void* state, void* stream) {
  VN_REQUIRE(state, "rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, ST, (uint32_t*)state);
  return vneti_check_launch("rng_advance");
}

extern "C" int vneti_sample_add_noise(const void* moments, long long ldm, const float* eps, const float* noise,
                                      const void* timesteps, const float* alphas_cumprod, float scaling,
                                      int v_prediction, float* latents, float* noisy, float* target, int Bn,
                                      int Lc, int HW, void* stream) {
  VN_REQUIRE(moments && eps && noise && timesteps && alphas_cumprod && latents && noisy && target,
             "sample_add_noise: null pointer");
  int n = Bn * Lc * HW;
  hipLaunchKernelGGL(sample_add_noise_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const half_t*)moments, ldm,
                     eps, noise, (const long long*)timesteps, alphas_cumprod, scaling, v_prediction, latents, noisy,
                     target, Bn, Lc, HW);
  return vneti_check_launch("sample_add_noise");
}

extern "C" int vneti_latent_sample(const void* moments, long long ldm, const float* eps, float scaling, float* latents,
                                   int Bn, int Lc, int HW, void* stream) {
  VN_REQUIRE(moments && eps && latents && Bn > 0 && Lc > 0 && HW > 0, "latent_sample: bad arguments");
  int n = Bn * Lc * HW;
  hipLaunchKernelGGL(latent_sample_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const half_t*)moments, ldm, eps, scaling,
                     latents, Bn, Lc, HW);
  return vneti_check_launch("latent_sample");
}

extern "C" int vneti_add_noise(const float* latents, const float* noise, const void* timesteps_i64,
                               const float* alphas_cumprod, int v_prediction, float* noisy, float* target, int Bn, int Lc,
                               int HW, void* stream) {
  VN_REQUIRE(latents && noise && timesteps_i64 && alphas_cumprod && (noisy || target) && Bn > 0 && Lc > 0 && HW > 0,
             "add_noise: bad arguments");
  int n = Bn * Lc * HW;
  hipLaunchKernelGGL(add_noise_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, latents, noise,
                     (const long long*)timesteps_i64, alphas_cumprod, v_prediction, noisy, target, Bn, Lc, HW);
  return vneti_check_launch("add_noise");
}

extern "C" int vneti_cfg_sampler_step(const void* pred, long long ldp, float* x, float* m_prev, float* x_in, int Bn,
                                      int Lc, int HW, float guidance, float alpha_t, float sigma_t, float cx, float c0,
                                      float c1, int v_prediction, void* stream) {
  VN_REQUIRE(pred && x && m_prev && x_in && Bn > 0 && Lc > 0 && HW > 0 && alpha_t > 0.f,
             "cfg_sampler_step: bad arguments");
  int n = Bn * Lc * HW;
  hipLaunchKernelGGL(cfg_sampler_step_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const half_t*)pred, ldp, x,
                     m_prev, x_in, Bn, Lc, HW, guidance, alpha_t, sigma_t, cx, c0, c1, v_prediction);
  return vneti_check_launch("cfg_sampler_step");
}

extern "C" int vneti_cfg_sampler_step_table(const void* pred, long long ldp, float* x, float* m_prev, float* x_in,
                                            int Bn, int Lc, int HW, float guidance, const float* coef_table,
                                            const int* step, int v_prediction, void* stream) {
  VN_REQUIRE(pred && x && m_prev && x_in && coef_table && step && Bn > 0 && Lc > 0 && HW > 0,
             "cfg_sampler_step_table: bad arguments");
  int n = Bn * Lc * HW;
  hipLaunchKernelGGL(cfg_sampler_step_table_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const half_t*)pred, ldp, x,
                     m_prev, x_in, Bn, Lc, HW, guidance, coef_table, step, v_prediction);
  return vneti_check_launch("cfg_sampler_step_table");
}

extern "C" int vneti_table_fill_i64(void* dst, int n, const void* table, const int* step, void* stream) {
  VN_REQUIRE(dst && table && step && n > 0, "table_fill_i64: bad arguments");
  hipLaunchKernelGGL(table_fill_i64_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (long long*)dst, n,
                     (const long long*)table, step);
  return vneti_check_launch("table_fill_i64");
}

extern "C" int vneti_counter_advance(int* counter, void* stream) {
  VN_REQUIRE(counter, "counter_advance: null pointer");
  hipLaunchKernelGGL(counter_advance_kernel, dim3(1), dim3(64), 0, ST, counter);
  return vneti_check_launch("counter_advance");
}

extern "C" int vneti_conv1x1_nchw_f32(const float* x, const float* W, const float* bias, float* out, int Bn, int Ci,
                                      int Co, int HW, float in_scale, void* stream) {
  VN_REQUIRE(x && W && out && Bn > 0 && Ci > 0 && Ci <= 8 && Co > 0 && Co <= 8 && HW > 0, "conv1x1_nchw: bad arguments");
  int n = Bn * Co * HW;
  hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, x, W, bias, out, Bn, Ci, Co, HW,
                     in_scale);
  return vneti_check_launch("conv1x1_nchw");
}

extern "C" int vneti_image_postprocess(const void* img, long long ldi, float* out, long long n_pix, int channels,
                                       void* stream) {
  VN_REQUIRE(img && out && n_pix > 0 && channels > 0 && channels <= ldi, "image_postprocess: bad arguments");
  hipLaunchKernelGGL(image_postprocess_kernel, dim3((unsigned)cdivl(n_pix * channels, 256)), dim3(256), 0, ST,
                     (const half_t*)img, ldi, out, n_pix, channels);
  return vneti_check_launch("image_postprocess");
}

extern "C" int vneti_mse_loss_grad(const void* pred, long long ldp, const float* target, void* dpred,
                                   long long lddp, float* loss_sum, const float* loss_scale, int Bn, int Lc,
                                   int HW, void* stream) {
  VN_REQUIRE(pred && target && dpred && loss_sum && loss_scale, "mse_loss_grad: null pointer");
  int n = Bn * Lc * HW;
  hipLaunchKernelGGL(mse_loss_grad_kernel, dim3(cdiv(n, 256)), dim3(256), 0, ST, (const half_t*)pred, ldp, target,
                     (half_t*)dpred, lddp, loss_sum, loss_scale, Bn, Lc, HW);
  return vneti_check_launch("mse_loss_grad");
}

extern "C" int vneti_adamw_segments(float* p, const float* g, float* m, float* v, long long seg_len, int n_seg,
                                    int* seg_step, const int* active, int n_active, const float* hyper,
                                    float* scaler, int* step, int growth_interval, int phases, void* stream) {
  VN_REQUIRE(p && g && m && v && hyper && scaler && step && seg_step && active && seg_len > 0 && n_seg > 0 &&
                 n_active > 0 && n_active <= 8 && phases > 0 && phases < 8,
             "adamw_segments: bad arguments");
  const long long n = seg_len * n_seg;
  if (phases & 1)
    hipLaunchKernelGGL(grads_check_finite_segments_kernel, dim3((unsigned)cdivl(seg_len, 256), n_active), dim3(256),
                       0, ST, g, seg_len, active, scaler);
  if (phases & 2) {
    hipLaunchKernelGGL(adamw_segments_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, ST, p, g, m, v, n, seg_len,
                       (const int*)seg_step, active, n_active, hyper, (const float*)scaler);
    hipLaunchKernelGGL(segments_step_kernel, dim3(cdiv(n_seg, 256)), dim3(256), 0, ST, seg_step, n_seg, active,
                       n_active, (const float*)scaler);
  }
  if (phases & 4) hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, ST, scaler, step, growth_interval);
  return vneti_check_launch("adamw_segments");
}

extern "C" int vneti_adamw_flat(float* p, const float* g, float* m, float* v, long long n, const float* hyper,
                                float* scaler, int* step, int growth_interval, int phases, void* stream) {
  VN_REQUIRE(p && g && m && v && hyper && scaler && step && n > 0 && phases > 0 && phases < 8,
             "adamw_flat: bad arguments");
  unsigned blocks = (unsigned)cdivl(n, 256);
  if (phases & 1) hipLaunchKernelGGL(grads_check_finite_kernel, dim3(blocks), dim3(256), 0, ST, g, n, scaler);
  if (phases & 2)
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, ST, p, g, m, v, n, hyper, (const float*)scaler,
                       (const int*)step);
  if (phases & 4) hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, ST, scaler, step, growth_interval);
  return vneti_check_launch("adamw_flat");
}
