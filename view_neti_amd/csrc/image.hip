// Device-side input pipeline: the per-sample image work of the reference's dataset --
// PIL bicubic/bilinear resize, torchvision ColorJitter / RandomGrayscale / GaussianBlur / RandomRotation /
// RandomResizedCrop (training/dataset.py:238-316, 700-740) -- on uint8 HWC RGB images resident in HBM.
// Every kernel restates the integer / float arithmetic of the library routine it replaces (Pillow
// src/libImaging: Resample.c, Geometry.c affine_fixed, Blend.c, Convert.c rgb2hsv/hsv2rgb, the L conversion;
// torchvision functional_tensor gaussian_blur as restated in compat/augment.py) so that a seeded run produces the
// same pixels as the host path; the random parameters are drawn on the host (compat/augment.py::draw_plan).
// These are tiny latency-bound launches (<= 768x576x3 bytes per image): one thread per pixel or per output sample.
#include "common.h"
#include "../../include/vneti.h"

// the restated routines round through specific float / double operations: no fused multiply-add contraction
#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c

__device__ __forceinline__ unsigned char clip8_fixed(int v) {  // Resample.c clip8: (in >> PRECISION_BITS) clamped
  v >>= PRECISION_BITS;
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__device__ __forceinline__ double bicubic_filter(double x) {  // Resample.c bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
__device__ __forceinline__ double bilinear_filter(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for one axis: one thread per output index.
// bounds[2*xx] = first source index, bounds[2*xx+1] = tap count; kk[xx*ksize + x] fixed-point weights.
__global__ void resample_coeffs_kernel(int in_size, int out_size, int filter, int ksize, int* bounds, int* kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double fsupport = filter == 0 ? 2.0 : 1.0;
  double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = fsupport * filterscale;
  const double center = (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  double w[64];
  for (int x = 0; x < xmax; ++x) {
    const double arg = (x + xmin - center + 0.5) * ss;
    w[x] = filter == 0 ? bicubic_filter(arg) : bilinear_filter(arg);
    ww += w[x];
  }
  int* k = kk + (long long)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    double v = 0.0;
    if (x < xmax) v = ww != 0.0 ? w[x] / ww : w[x];
    k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

// Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc, 3 interleaved channels.
// HORIZ: out[y][xx] = sum_x in[y][xmin+x] * k[x];  else out[yy][x] = sum_y in[ymin+y][x] * k[y]
template <bool HORIZ>
__global__ void resample_pass_kernel(const unsigned char* in, int in_w, unsigned char* out, int out_h, int out_w,
                                     const int* bounds, const int* kk, int ksize) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)out_h * out_w) return;
  const int y = (int)(gid / out_w), x = (int)(gid % out_w);
  const int o = HORIZ ? x : y;
  const int lo = bounds[2 * o], n = bounds[2 * o + 1];
  const int* k = kk + (long long)o * ksize;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int t = 0; t < n; ++t) {
    const unsigned char* p = HORIZ ? in + ((long long)y * in_w + lo + t) * 3 : in + ((long long)(lo + t) * in_w + x) * 3;
    s0 += p[0] * k[t];
    s1 += p[1] * k[t];
    s2 += p[2] * k[t];
  }
  unsigned char* q = out + ((long long)y * out_w + x) * 3;
  q[0] = clip8_fixed(s0);
  q[1] = clip8_fixed(s1);
  q[2] = clip8_fixed(s2);
}

// crop (+ optional horizontal flip = Image.transpose(FLIP_LEFT_RIGHT) of the cropped image)
__global__ void crop_kernel(const unsigned char* in, int in_w, int top, int left, unsigned char* out, int h, int w,
                            int flip) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)h * w) return;
  const int y = (int)(gid / w), x = (int)(gid % w);
  const int sx = flip ? (w - 1 - x) : x;
  const unsigned char* p = in + ((long long)(top + y) * in_w + left + sx) * 3;
  unsigned char* q = out + gid * 3;
  q[0] = p[0];
  q[1] = p[1];
  q[2] = p[2];
}

__device__ __forceinline__ int luma(const unsigned char* p) {  // Convert.c L24: ITU-R 601-2 in 16.16 fixed point
  return (p[0] * 19595 + p[1] * 38470 + p[2] * 7471 + 0x8000) >> 16;
}

// sum of the L conversion over the image (ImageStat.Stat(im.convert("L")).sum[0]) -> acc[0] (64-bit)
__global__ void luma_sum_kernel(const unsigned char* img, long long n, unsigned long long* acc) {
  __shared__ unsigned long long part[256];
  unsigned long long s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    s += (unsigned long long)luma(img + i * 3);
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(acc, part[0]);
}

// Blend.c ImagingBlend(degenerate, image, alpha) as used by ImageEnhance: out = deg + alpha * (img - deg).
// MODE 0: degenerate = constant 0 (Brightness); 1: constant int(mean(L) + 0.5) read from the luma sum (Contrast);
// 2: per-pixel L (Color); 3: RandomGrayscale / to_grayscale3 (out = L, alpha unused)
template <int MODE>
__global__ void enhance_kernel(unsigned char* img, long long n, float alpha, const unsigned long long* lsum) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  unsigned char* p = img + gid * 3;
  int deg = 0;
  if (MODE == 1) deg = (int)((double)lsum[0] / (double)n + 0.5);
  if (MODE >= 2) deg = luma(p);
  if (MODE == 3) {
    p[0] = p[1] = p[2] = (unsigned char)deg;
    return;
  }
  const bool inside = alpha >= 0.f && alpha <= 1.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t = (float)deg + alpha * (float)((int)p[c] - deg);
    if (inside) {
      p[c] = (unsigned char)(int)t;
    } else {
      p[c] = t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)(int)t);
    }
  }
}

// adjust_hue (compat/augment.py, torchvision functional_pil): RGB -> HSV (Convert.c rgb2hsv_row), H += shift (uint8
// wrap), HSV -> RGB (Convert.c hsv2rgb)
__global__ void hue_kernel(unsigned char* img, long long n, int shift) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  unsigned char* px = img + gid * 3;
  const unsigned char r = px[0], g = px[1], b = px[2];
  const unsigned char maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  unsigned char uh = 0, us = 0;
  const unsigned char uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + rc - bc);
    else h = (float)(4.0 + gc - rc);
    h = (float)fmod((h / 6.0 + 1.0), 1.0);
    int ih = (int)(h * 255.0), is = (int)(s * 255.0);
    uh = (unsigned char)(ih < 0 ? 0 : (ih > 255 ? 255 : ih));
    us = (unsigned char)(is < 0 ? 0 : (is > 255 ? 255 : is));
  }
  uh = (unsigned char)((int)uh + shift);  // np.uint8 wrap-around
  if (us == 0) {
    px[0] = px[1] = px[2] = uv;
    return;
  }
  const int i = (int)floor((float)uh * 6.0 / 255.0);
  const float f = (float)((float)uh * 6.0 / 255.0 - (float)i);
  const float fs = (float)(((float)us) / 255.0);
  int p = (int)round((float)uv * (1.0 - fs));
  int q = (int)round((float)uv * (1.0 - fs * f));
  int t = (int)round((float)uv * (1.0 - fs * (1.0 - f)));
  const unsigned char up = (unsigned char)(p < 0 ? 0 : (p > 255 ? 255 : p));
  const unsigned char uq = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
  const unsigned char ut = (unsigned char)(t < 0 ? 0 : (t > 255 ? 255 : t));
  switch (i % 6) {
    case 0: px[0] = uv; px[1] = ut; px[2] = up; break;
    case 1: px[0] = uq; px[1] = uv; px[2] = up; break;
    case 2: px[0] = up; px[1] = uv; px[2] = ut; break;
    case 3: px[0] = up; px[1] = uq; px[2] = uv; break;
    case 4: px[0] = ut; px[1] = up; px[2] = uv; break;
    default: px[0] = uv; px[1] = up; px[2] = uq; break;
  }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {  // np.pad(mode="reflect")
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}
// 5-tap separable Gaussian in f32 with reflect padding (compat/augment.py::blur_with_sigma): the horizontal pass
// keeps f32 (tmp), the vertical pass rounds half-to-even and clips.  Accumulation order = Python's sum(): left to right
// starting from 0.
struct Blur5 {
  float k[5];
};
__global__ void blur_h_kernel(const unsigned char* in, float* tmp, int h, int w, Blur5 kw) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)h * w) return;
  const int y = (int)(gid / w), x = (int)(gid % w);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) acc = acc + kw.k[i] * (float)in[((long long)y * w + reflect_idx(x + i - 2, w)) * 3 + c];
    tmp[gid * 3 + c] = acc;
  }
}
__global__ void blur_v_kernel(const float* tmp, unsigned char* out, int h, int w, Blur5 kw) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)h * w) return;
  const int y = (int)(gid / w), x = (int)(gid % w);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) acc = acc + kw.k[i] * tmp[((long long)reflect_idx(y + i - 2, h) * w + x) * 3 + c];
    float r = rintf(acc);
    r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
    out[gid * 3 + c] = (unsigned char)r;
  }
}

// Geometry.c affine_fixed (nearest, 16.16 fixed point): out(x, y) = in(xx >> 16, yy >> 16) where inside, else `fill`
__global__ void affine_nearest_kernel(const unsigned char* in, unsigned char* out, int h, int w, int a0, int a1, int a2,
                                      int a3, int a4, int a5, int fill) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)h * w) return;
  const int y = (int)(gid / w), x = (int)(gid % w);
  const int xx = a2 + y * a1 + x * a0, yy = a5 + y * a4 + x * a3;
  const int xin = xx >> 16, yin = yy >> 16;
  unsigned char* q = out + gid * 3;
  if (xin >= 0 && xin < w && yin >= 0 && yin < h) {
    const unsigned char* p = in + ((long long)yin * w + xin) * 3;
    q[0] = p[0];
    q[1] = p[1];
    q[2] = p[2];
  } else {
    q[0] = q[1] = q[2] = (unsigned char)fill;
  }
}

// dataset.py:738-739: (uint8 / 127.5 - 1.0) in float64, cast to f32, HWC -> CHW
__global__ void to_f32_chw_kernel(const unsigned char* img, float* out, int h, int w) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)h * w;
  if (gid >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[c * n + gid] = (float)((double)img[gid * 3 + c] / 127.5 - 1.0);
}

inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int vneti_img_resample_ksize(int in_size, int out_size, int filter) {
  double scale = (double)in_size / out_size;
  if (scale < 1.0) scale = 1.0;
  const double support = (filter == 0 ? 2.0 : 1.0) * scale;
  return (int)ceil(support) * 2 + 1;
}

extern "C" int vneti_img_resample_coeffs(int in_size, int out_size, int filter, int* bounds, int* kk, void* stream) {
  VN_REQUIRE(in_size > 0 && out_size > 0 && (filter == 0 || filter == 1) && bounds && kk, "img_resample_coeffs: bad arguments");
  const int ksize = vneti_img_resample_ksize(in_size, out_size, filter);
  VN_REQUIRE(ksize <= 64, "img_resample_coeffs: scale too large (ksize %d > 64)", ksize);
  hipLaunchKernelGGL(resample_coeffs_kernel, dim3(blocks_for(out_size)), dim3(256), 0, ST, in_size, out_size, filter,
                     ksize, bounds, kk);
  return vneti_check_launch("img_resample_coeffs");
}

extern "C" int vneti_img_resample_pass(const void* in, int in_w, void* out, int out_h, int out_w, const int* bounds,
                                       const int* kk, int ksize, int horizontal, void* stream) {
  VN_REQUIRE(in && out && bounds && kk && out_h > 0 && out_w > 0 && ksize > 0, "img_resample_pass: bad arguments");
  const long long n = (long long)out_h * out_w;
  if (horizontal)
    hipLaunchKernelGGL(resample_pass_kernel<true>, dim3(blocks_for(n)), dim3(256), 0, ST, (const unsigned char*)in, in_w,
                       (unsigned char*)out, out_h, out_w, bounds, kk, ksize);
  else
    hipLaunchKernelGGL(resample_pass_kernel<false>, dim3(blocks_for(n)), dim3(256), 0, ST, (const unsigned char*)in, in_w,
                       (unsigned char*)out, out_h, out_w, bounds, kk, ksize);
  return vneti_check_launch("img_resample_pass");
}

extern "C" int vneti_img_crop(const void* in, int in_w, int top, int left, void* out, int h, int w, int flip, void* stream) {
  VN_REQUIRE(in && out && h > 0 && w > 0 && top >= 0 && left >= 0 && left + w <= in_w, "img_crop: bad arguments");
  hipLaunchKernelGGL(crop_kernel, dim3(blocks_for((long long)h * w)), dim3(256), 0, ST, (const unsigned char*)in, in_w, top,
                     left, (unsigned char*)out, h, w, flip);
  return vneti_check_launch("img_crop");
}

extern "C" int vneti_img_enhance(void* img, int h, int w, int mode, float alpha, void* scratch8, void* stream) {
  VN_REQUIRE(img && h > 0 && w > 0 && mode >= 0 && mode <= 3 && (mode != 1 || scratch8), "img_enhance: bad arguments");
  const long long n = (long long)h * w;
  unsigned long long* acc = (unsigned long long*)scratch8;
  unsigned char* p = (unsigned char*)img;
  switch (mode) {
    case 0: hipLaunchKernelGGL(enhance_kernel<0>, dim3(blocks_for(n)), dim3(256), 0, ST, p, n, alpha, acc); break;
    case 1:
      (void)hipMemsetAsync(acc, 0, 8, ST);
      hipLaunchKernelGGL(luma_sum_kernel, dim3(min(blocks_for(n), 256u)), dim3(256), 0, ST, (const unsigned char*)p, n, acc);
      hipLaunchKernelGGL(enhance_kernel<1>, dim3(blocks_for(n)), dim3(256), 0, ST, p, n, alpha, acc);
      break;
    case 2: hipLaunchKernelGGL(enhance_kernel<2>, dim3(blocks_for(n)), dim3(256), 0, ST, p, n, alpha, acc); break;
    default: hipLaunchKernelGGL(enhance_kernel<3>, dim3(blocks_for(n)), dim3(256), 0, ST, p, n, alpha, acc); break;
  }
  return vneti_check_launch("img_enhance");
}

extern "C" int vneti_img_hue(void* img, int h, int w, int shift, void* stream) {
  VN_REQUIRE(img && h > 0 && w > 0, "img_hue: bad arguments");
  const long long n = (long long)h * w;
  hipLaunchKernelGGL(hue_kernel, dim3(blocks_for(n)), dim3(256), 0, ST, (unsigned char*)img, n, shift);
  return vneti_check_launch("img_hue");
}

extern "C" int vneti_img_blur5(const void* in, void* out, float* tmp, int h, int w, const float* k5_host, void* stream) {
  VN_REQUIRE(in && out && tmp && k5_host && h >= 3 && w >= 3, "img_blur5: bad arguments");
  Blur5 kw;
  for (int i = 0; i < 5; ++i) kw.k[i] = k5_host[i];
  const long long n = (long long)h * w;
  hipLaunchKernelGGL(blur_h_kernel, dim3(blocks_for(n)), dim3(256), 0, ST, (const unsigned char*)in, tmp, h, w, kw);
  hipLaunchKernelGGL(blur_v_kernel, dim3(blocks_for(n)), dim3(256), 0, ST, (const float*)tmp, (unsigned char*)out, h, w, kw);
  return vneti_check_launch("img_blur5");
}

extern "C" int vneti_img_affine_nearest(const void* in, void* out, int h, int w, const int* a6_host, int fill, void* stream) {
  VN_REQUIRE(in && out && in != out && a6_host && h > 0 && w > 0, "img_affine_nearest: bad arguments");
  hipLaunchKernelGGL(affine_nearest_kernel, dim3(blocks_for((long long)h * w)), dim3(256), 0, ST, (const unsigned char*)in,
                     (unsigned char*)out, h, w, a6_host[0], a6_host[1], a6_host[2], a6_host[3], a6_host[4], a6_host[5],
                     fill);
  return vneti_check_launch("img_affine_nearest");
}

extern "C" int vneti_img_to_f32_chw(const void* img, float* out, int h, int w, void* stream) {
  VN_REQUIRE(img && out && h > 0 && w > 0, "img_to_f32_chw: bad arguments");
  hipLaunchKernelGGL(to_f32_chw_kernel, dim3(blocks_for((long long)h * w)), dim3(256), 0, ST, (const unsigned char*)img,
                     out, h, w);
  return vneti_check_launch("img_to_f32_chw");
}
