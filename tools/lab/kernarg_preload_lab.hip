// Lab: what does a kernel pay for fetching its (cold) kernel arguments, and does gfx950's kernarg preload
// (-mllvm -amdgpu-kernarg-preload-count=16: the first 16 dwords of explicit scalar / pointer arguments arrive in SGPRs with
// the wave) take it off the launch?  A by-value struct argument is never preloaded, explicit arguments are.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_preload_lab kernarg_preload_lab.hip
// Chain per repetition: [evict: 640 MB fill] [small kernel], timed with events over 40 repetitions, against the same chain
// with an EMPTY kernel (no arguments used) — and the in-kernel s_memtime between entry and "arguments usable".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Args { const float* a; float* b; int n; int m; float s; int pad[27]; unsigned long long* stamp; };
__global__ __launch_bounds__(256) void k_struct(Args g) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int i = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (i < g.n) v = g.a[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (i < g.n) g.b[i] = v * g.s + g.m;
  if (threadIdx.x == 0 && g.stamp) { g.stamp[2 * blockIdx.x] = t0; g.stamp[2 * blockIdx.x + 1] = t1; }
}
__global__ __launch_bounds__(256) void k_args(const float* a, float* b, int n, int m, float s, unsigned long long* stamp) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int i = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (i < n) v = a[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (i < n) b[i] = v * s + m;
  if (threadIdx.x == 0 && stamp) { stamp[2 * blockIdx.x] = t0; stamp[2 * blockIdx.x + 1] = t1; }
}
__global__ __launch_bounds__(256) void k_fill(float4* p, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = float4{0, 0, 0, 0};
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int n = 256 * 256;
  float *a, *b; float4* big; unsigned long long* st;
  const size_t bigN = 640ull * 1024 * 1024 / 16;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&big, bigN * 16)); CK(hipMalloc(&st, 256 * 16));
  CK(hipMemset(a, 0, n * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int variant = 0; variant < 3; ++variant) {
    // one graph: (fill, small) x 20
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < 20; ++r) {
      hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, s, big, bigN);
      if (variant == 0) { Args g2{}; g2.a = a; g2.b = b; g2.n = n; g2.m = 1; g2.s = 2.f; g2.stamp = st; hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, s, g2); }
      else if (variant == 1) hipLaunchKernelGGL(k_args, dim3(256), dim3(256), 0, s, (const float*)a, b, n, 1, 2.f, st);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<double> d;
    for (int i = 0; i < 256; ++i) d.push_back((double)(h[2 * i + 1] - h[2 * i]));
    std::sort(d.begin(), d.end());
    printf("%-28s: graph of 20 x (fill + kernel): median %.3f ms => %.2f us per pair; in-kernel entry -> first data: median %.0f ticks (min %.0f)\n",
           variant == 0 ? "by-value struct (s_load)" : variant == 1 ? "explicit args (preloaded)" : "fill only", ts[ts.size() / 2], ts[ts.size() / 2] * 1e3 / 20,
           variant < 2 ? d[128] : 0.0, variant < 2 ? d[0] : 0.0);
  }
  return 0;
}
