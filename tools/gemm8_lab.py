"""Dev tool: timing of tile 16 (and 5) on a few shapes; run under VNETI_GEMM8_LAB / VNETI_GEMM8_PH to ablate the loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops
dev = "cuda"
HINTS = tuple(int(h) for h in os.environ.get("HINTS", "16").split(","))


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for M, N, K in [(4096, 4096, 64), (4096, 4096, 1024), (4096, 4096, 4096), (4096, 4096, 16384), (65536, 512, 4608)]:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    gf = 2.0 * M * N * K / 1e9
    out = []
    for h in HINTS:
        t = min(timeit(lambda: ops.gemm(A, B, C, tile_hint=h, split_k=1)) for _ in range(3))
        out.append(f"h{h} {t:8.1f}us {gf / t * 1e3:5.0f}TF")
    print(f"lab={os.environ.get('VNETI_GEMM8_LAB', '0')} ph={os.environ.get('VNETI_GEMM8_PH', '4')} M={M} N={N} K={K}: " + " ".join(out), flush=True)
