"""Lab (round 4; the tiles are NOT in the tree any more — git log: "160-wide tiles"): the 160-wide tiles (19 / 20 / 21: 64x160
ring4, 64x160 two-stage, 128x160 ring3; every SD-1.5 width is a multiple of 320, and 4096 x 640 is exactly 256 tiles of 64 x 160) against the 128- / 64-wide ones on the short-K linears of the step, cache-hot and
cold (a 640 MB fill between launches, as the autotuner times them)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
for (M, N, K) in [(4096, 640, 640), (16384, 320, 320), (1024, 1280, 1280), (16384, 320, 1280), (4096, 640, 2560), (16384, 1280, 320), (4096, 2560, 640)]:
    a = torch.randn(M, K, device=dev).half()
    b = (torch.randn(N, K, device=dev) * 0.03).half()
    c = torch.empty(M, N, device=dev, dtype=torch.float16)
    r = torch.randn(M, N, device=dev).half()
    bias = torch.randn(N, device=dev)
    out = []
    for h in (9, 13, 15, 19, 20, 21):
        f = lambda: ops.gemm(a, b, c, bias=bias, resid=r, tile_hint=h, workspace=ws, split_k=1)
        f(); f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record(); torch.cuda.synchronize()
        hot = s.elapsed_time(e) / 20 * 1e3
        ts = []
        for _ in range(7):
            cold.fill_(0); a.add_(0)
            s.record(); f(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        out.append(f"h{h} {hot:5.1f}/{sorted(ts)[3]:5.1f}")
    print(f"{M}x{N}x{K}: hot/cold us  " + "  ".join(out), flush=True)
