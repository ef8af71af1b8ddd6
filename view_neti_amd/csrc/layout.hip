// Layout helpers: batched f16 transpose (attention operands) and the small-Cin 3x3 im2col.
#include "common.h"
#include "../../include/vneti.h"

namespace {

// out[b][c][r] = in[b][r][c]; out columns r in [rows, ld_out) are zero filled.
__global__ __launch_bounds__(256) void transpose_kernel(const half_t* __restrict__ in, long long ld_in,
                                                        long long stride_in, half_t* __restrict__ out,
                                                        long long ld_out, long long stride_out, int rows,
                                                        int cols) {
  __shared__ half_t tile[64][66];
  const int b = blockIdx.z;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const half_t* ib = in + (long long)b * stride_in;
  half_t* ob = out + (long long)b * stride_out;
  for (int idx = threadIdx.x; idx < 512; idx += 256) {
    int r = idx >> 3, ch = idx & 7;
    int gr = r0 + r, gc = c0 + ch * 8;
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
    if (gr < rows && gc < cols) v = *reinterpret_cast<const half8*>(ib + (long long)gr * ld_in + gc);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[r][ch * 8 + j] = v[j];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 512; idx += 256) {
    int c = idx & 63, rch = idx >> 6;  // lanes walk columns -> conflict-free LDS column reads
    int gc = c0 + c, gr = r0 + rch * 8;
    if (gc < cols && gr < ld_out) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[rch * 8 + j][c];
      *reinterpret_cast<half8*>(ob + (long long)gc * ld_out + gr) = v;
    }
  }
}

struct TMulti {
  vneti_transpose_desc d[VNETI_TRANSPOSE_MAX];
  int n;
};

// several independent transposes in one grid: blockIdx.z walks the concatenated batches
__global__ __launch_bounds__(256) void transpose_multi_kernel(TMulti m) {
  __shared__ half_t tile[64][66];
  int z = blockIdx.z, k = 0;
  while (k < m.n - 1 && z >= m.d[k].batch) {
    z -= m.d[k].batch;
    ++k;
  }
  const vneti_transpose_desc& d = m.d[k];
  const int rows = d.rows, cols = d.cols;
  const long long ld_in = d.ld_in, ld_out = d.ld_out;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  if (r0 >= ld_out || c0 >= cols) return;  // grid is sized for the largest descriptor
  const half_t* ib = (const half_t*)d.in + (long long)z * d.stride_in;
  half_t* ob = (half_t*)d.out + (long long)z * d.stride_out;
  for (int idx = threadIdx.x; idx < 512; idx += 256) {
    int r = idx >> 3, ch = idx & 7;
    int gr = r0 + r, gc = c0 + ch * 8;
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
    if (gr < rows && gc < cols) v = *reinterpret_cast<const half8*>(ib + (long long)gr * ld_in + gc);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[r][ch * 8 + j] = v[j];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 512; idx += 256) {
    int c = idx & 63, rch = idx >> 6;
    int gc = c0 + c, gr = r0 + rch * 8;
    if (gc < cols && gr < ld_out) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[rch * 8 + j][c];
      *reinterpret_cast<half8*>(ob + (long long)gc * ld_out + gr) = v;
    }
  }
}

// out[m][k], k = tap*C + c for tap<9, c<C; zero for k >= 9*C (row length 64).
template <bool XF32>
__global__ __launch_bounds__(256) void im2col_small_kernel(const void* __restrict__ x, long long sb, long long sc,
                                                           long long sy, long long sx, half_t* __restrict__ out,
                                                           int M, int C, int Hi, int Wi, int Ho, int Wo,
                                                           int stride, int pad_t, int pad_l) {
  long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  long long m = gid >> 3;
  int ch = (int)(gid & 7);
  if (m >= M) return;
  int hw = Ho * Wo;
  int b = (int)(m / hw);
  int rem = (int)(m - (long long)b * hw);
  int oy = rem / Wo, ox = rem - oy * Wo;
  half8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int k = ch * 8 + j;
    float val = 0.f;
    if (k < 9 * C) {
      int tap = k / C, c = k - tap * C;
      int dy = tap / 3, dx = tap - dy * 3;
      int iy = oy * stride + dy - pad_t, ix = ox * stride + dx - pad_l;
      if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
        long long off = (long long)b * sb + (long long)c * sc + (long long)iy * sy + (long long)ix * sx;
        val = XF32 ? reinterpret_cast<const float*>(x)[off] : (float)reinterpret_cast<const half_t*>(x)[off];
      }
    }
    v[j] = (half_t)val;
  }
  *reinterpret_cast<half8*>(out + m * 64 + ch * 8) = v;
}

}  // namespace

extern "C" int vneti_transpose_f16(const void* in, long long ld_in, long long stride_in, void* out,
                                   long long ld_out, long long stride_out, int rows, int cols, int batch,
                                   void* stream) {
  VN_REQUIRE(in && out && rows > 0 && cols > 0 && batch > 0, "transpose: bad arguments");
  VN_REQUIRE(cols % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ld_out >= rows,
             "transpose: cols/ld must be multiples of 8 and ld_out >= rows");
  dim3 grid(cdiv((int)ld_out, 64), cdiv(cols, 64), batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const half_t*)in, ld_in,
                     stride_in, (half_t*)out, ld_out, stride_out, rows, cols);
  return vneti_check_launch("transpose");
}

extern "C" int vneti_transpose_f16_multi(const vneti_transpose_desc* descs, int n, void* stream) {
  VN_REQUIRE(descs && n >= 1 && n <= VNETI_TRANSPOSE_MAX, "transpose_multi: need 1..%d descriptors", VNETI_TRANSPOSE_MAX);
  TMulti m;
  m.n = n;
  int gx = 0, gy = 0, gz = 0;
  for (int i = 0; i < n; ++i) {
    const vneti_transpose_desc& d = descs[i];
    VN_REQUIRE(d.in && d.out && d.rows > 0 && d.cols > 0 && d.batch > 0, "transpose_multi[%d]: bad arguments", i);
    VN_REQUIRE(d.cols % 8 == 0 && d.ld_in % 8 == 0 && d.ld_out % 8 == 0 && d.ld_out >= d.rows,
               "transpose_multi[%d]: cols/ld must be multiples of 8 and ld_out >= rows", i);
    m.d[i] = d;
    gx = gx > cdiv((int)d.ld_out, 64) ? gx : cdiv((int)d.ld_out, 64);
    gy = gy > cdiv(d.cols, 64) ? gy : cdiv(d.cols, 64);
    gz += d.batch;
  }
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(gx, gy, gz), dim3(256), 0, (hipStream_t)stream, m);
  return vneti_check_launch("transpose_multi");
}

extern "C" int vneti_im2col3x3_small(const void* x, int x_is_f32, long long sb, long long sc, long long sy,
                                     long long sx, void* out, int Bn, int C, int Hi, int Wi, int Ho, int Wo,
                                     int stride, int pad_t, int pad_l, void* stream) {
  VN_REQUIRE(x && out && Bn > 0 && C > 0 && 9 * C <= 64, "im2col_small: need 9*C <= 64 (C=%d)", C);
  long long M = (long long)Bn * Ho * Wo;
  VN_REQUIRE(M < 0x7fffffffLL / 8, "im2col_small: too many rows");
  dim3 grid((unsigned)cdivl(M * 8, 256));
  if (x_is_f32)
    hipLaunchKernelGGL((im2col_small_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, sb, sc, sy, sx,
                       (half_t*)out, (int)M, C, Hi, Wi, Ho, Wo, stride, pad_t, pad_l);
  else
    hipLaunchKernelGGL((im2col_small_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, sb, sc, sy, sx,
                       (half_t*)out, (int)M, C, Hi, Wi, Ho, Wo, stride, pad_t, pad_l);
  return vneti_check_launch("im2col_small");
}
