"""Lab: the UNet's small GEMMs under the LDS-DMA tiles against the register-staged variants (tile_hint + 100: buffer_load to
VGPRs + ds_write, two stages) — an LDS-DMA blocks its wave for 60-185 cycles per instruction, and the 64x64 / 128x64 tiles run
one wave per SIMD.  Cold timings (640 MB fill, activation re-touched), medians of 9, events around single launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
for (M, N, K) in [(1024, 1280, 1280), (4096, 640, 640), (16384, 320, 320), (256, 1280, 1280), (1024, 1280, 5120), (4096, 320, 640), (308, 640, 768)]:
    a = torch.randn(M, K, device=dev).half()
    b = (torch.randn(N, K, device=dev) * 0.03).half()
    c = torch.empty(M, N, device=dev, dtype=torch.float16)
    r = torch.randn(M, N, device=dev).half()
    bias = torch.randn(N, device=dev)
    out = []
    for h in (3, 12, 15, 103, 2, 14, 102, 1, 13, 101):
        f = lambda: ops.gemm(a, b, c, bias=bias, resid=r, tile_hint=h, workspace=ws, split_k=1)
        f(); f()
        ts = []
        for _ in range(9):
            cold.fill_(0); a.add_(0); r.add_(0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); f(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        out.append(f"h{h} {sorted(ts)[4]:5.1f}")
    print(f"{M}x{N}x{K}: cold us  " + "  ".join(out), flush=True)
