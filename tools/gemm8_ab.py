"""Dev tool: the 8-phase 256x256 tile (tile_hint 16, csrc/gemm8.hip) against the generic 256x256 tile (5), the 256x128
ring tiles (6, 7) and hipBLASLt (torch.matmul) — correctness first (bit-level agreement is not expected: different
accumulation order; compared against an fp32 torch reference, repeated to screen for races), then time on random f16
operands (power-capped clocks: quote these, not constant-operand numbers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops

dev = "cuda"
HINTS = tuple(int(h) for h in os.environ.get("HINTS", "5,16,7,17").split(","))
ops.set_default_gemm_workspace(torch.empty(64 * 2 ** 20, dtype=torch.float32, device=dev))


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


# ---- correctness + race screen ----
bad = 0
for (M, N, K) in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 512), (512, 512, 4096), (300, 320, 128),
                  (1000, 264, 320), (4096, 4096, 4096), (4928, 3072, 768), (16384, 320, 2880)]:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).half().to(dev)
    B = torch.randn(N, K, generator=g).half().to(dev)
    ref = (A.float() @ B.float().t())
    for TH in (16, 17):
        for rep in range(5):
            C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
            ops.gemm(A, B, C, tile_hint=TH)
            torch.cuda.synchronize()
            err = ((C.float() - ref).norm() / ref.norm()).item()
            mx = (C.float() - ref).abs().max().item()
            if not (err < 2e-3):
                bad += 1
                print(f"MISMATCH tile {TH} M={M} N={N} K={K} rep {rep}: rel {err:.3e} max {mx:.3e} nan {int(torch.isnan(C).sum())}")
                break
        else:
            print(f"ok tile {TH} M={M} N={N} K={K}: rel {err:.2e} max {mx:.2e} (5 runs)")
        # split-K through the same kernel
        if K >= 512:
            C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
            ops.gemm(A, B, C, tile_hint=TH, split_k=3)
            torch.cuda.synchronize()
            err = ((C.float() - ref).norm() / ref.norm()).item()
            print(f"   tile {TH} split_k=3: rel {err:.2e}")
            bad += not (err < 2e-3)
print("CORRECTNESS", "FAILED" if bad else "OK")

# ---- time ----
SHAPES = [(4096, 4096, 4096), (8192, 8192, 4096), (262144, 256, 2304), (65536, 512, 4608), (16384, 512, 4608),
          (16384, 640, 5760), (16384, 320, 2880), (4096, 1280, 11520), (4096, 640, 5760),
          (16384, 2560, 320), (4096, 5120, 640), (1024, 10240, 1280), (4928, 2304, 768), (4928, 3072, 768), (4928, 768, 3072)]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    Bt = B.t()
    gf = 2.0 * M * N * K / 1e9
    t_blas = timeit(lambda: torch.matmul(A, Bt, out=C))
    # interleaved A/B: two rounds per hint
    ts = {h: [] for h in HINTS}
    for _ in range(2):
        for h in HINTS:
            ts[h].append(timeit(lambda: ops.gemm(A, B, C, tile_hint=h, split_k=1)))
    line = " ".join(f"h{h} {min(v):8.1f}us {gf / min(v) * 1e3:5.0f}TF" for h, v in ts.items())
    print(f"M={M:7d} N={N:5d} K={K:5d} {gf:8.1f}GF | blas {t_blas:8.1f}us {gf / t_blas * 1e3:5.0f}TF | {line}", flush=True)
    del A, B, C
