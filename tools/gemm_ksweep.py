"""Dev tool: per-tile fixed cost vs per-k-step cost of each GEMM tile configuration (fit t = a + b*ksteps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops

dev = "cuda"
TILES = {1: (128, 128), 2: (128, 64), 3: (64, 64), 4: (256, 128), 5: (256, 256), 6: (256, 128), 7: (256, 128), 8: (256, 128), 9: (128, 128), 10: (128, 128), 11: (128, 64), 12: (64, 64)}
M = int(os.environ.get("M", 65536)); N = int(os.environ.get("N", 256))
for hint in [int(x) for x in os.environ.get("HINTS", "5,4,7,8,1,9,2,3").split(",")]:
    bm, bn = TILES[hint]
    rounds = (M // bm) * (N // bn) / 256.0
    res = []
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        if os.environ.get("DATA") == "const":
            A = torch.full((M, K), 1.0, device=dev, dtype=torch.float16); B = torch.full((N, K), 1.0, device=dev, dtype=torch.float16)
        elif os.environ.get("DATA") == "small":   # few distinct values, random signs: low toggle rate
            A = torch.randint(0, 2, (M, K), device=dev).to(torch.float16); B = torch.randint(0, 2, (N, K), device=dev).to(torch.float16)
        else:
            A = torch.randn(M, K, device=dev, dtype=torch.float16); B = torch.randn(N, K, device=dev, dtype=torch.float16)
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        f = lambda: ops.gemm(A, B, out, tile_hint=hint, split_k=1)
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        res.append((K // 64, s.elapsed_time(e) / 10 * 1e3))
    (k0, t0), (k1, t1) = res[-3], res[-1]
    b = (t1 - t0) / (k1 - k0); a = t1 - b * k1
    ideal = bm * bn * 64 * 2 / (2.5e15 / 256) * 1e6
    print(f"tile {hint} {bm}x{bn}: rounds={rounds:.1f} " + " ".join(f"k{k}:{t:.1f}" for k, t in res) +
          f" | per-round fixed {a/rounds:.2f}us, per-kstep {b/rounds:.3f}us (ideal {ideal:.3f})")
