"""Lab: which Tensile kernels does hipBLASLt pick for the step's plain-GEMM shapes (their names encode macro tile,
MFMA shape, wave layout, LDS options)?  Run under rocprofv3 --kernel-trace --stats."""
import torch
shapes = [(4096, 4096, 4096), (16384, 640, 5760), (4928, 3072, 768), (4928, 768, 3072), (16384, 320, 2880), (65536, 512, 4608),
          (16384, 320, 320), (4096, 640, 640), (1024, 1280, 11520)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").half(); b = torch.randn(N, K, device="cuda").half()
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): c = a @ b.t()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    print(f"{M}x{N}x{K}: {t*1e3:.1f} us {2*M*N*K/t/1e9:.0f} TF/s", flush=True)
