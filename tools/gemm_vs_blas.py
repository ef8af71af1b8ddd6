"""Dev tool: yardstick — time torch.matmul (hipBLASLt/rocBLAS f16) on the plain-GEMM equivalents of the step's
GEMM/conv problems beside this repo's kernel (best tile config), to see the headroom per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops

dev = "cuda"
HINTS = tuple(int(h) for h in os.environ.get("HINTS", "1,2,4,5").split(","))
ops.set_default_gemm_workspace(torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev))
SHAPES = [
    (4096, 4096, 4096),
    (524288, 128, 1152), (262144, 256, 2304), (65536, 512, 4608),   # VAE convs as GEMMs
    (16384, 320, 2880), (4096, 640, 5760), (1024, 1280, 11520), (256, 1280, 11520),
    (16384, 640, 5760), (4096, 1280, 11520), (1024, 2560, 11520),
    (16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280),
    (16384, 2560, 320), (4096, 5120, 640), (1024, 10240, 1280),
    (16384, 320, 1280), (4096, 640, 2560), (1024, 1280, 5120),
    (4928, 768, 768), (4928, 2304, 768), (4928, 3072, 768), (4928, 768, 3072),
]


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    Bt = B.t()
    t_blas = timeit(lambda: torch.matmul(A, Bt, out=C))
    ts = {h: timeit(lambda: ops.gemm(A, B, C, tile_hint=h)) for h in HINTS}
    best = min(ts, key=ts.get)
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:8d} N={N:6d} K={K:6d} {gf:9.1f}GF | blas {t_blas:9.1f}us {gf / t_blas * 1e3:6.0f}TF | ours h{best} "
          f"{ts[best]:9.1f}us {gf / ts[best] * 1e3:6.0f}TF | ratio {t_blas / ts[best]:.2f}", flush=True)
    del A, B, C
