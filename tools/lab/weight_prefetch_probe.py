"""Dev tool (GPU box): how much of a GEMM launch of the step is the first touch of its (always cold) weights, and does a
read of the weights a little EARLIER — by another kernel, on any XCD — take that off the launch?  The MALL (256 MB
Infinity Cache) is memory-side: a line read through one XCD's L2 is a MALL hit for every other XCD afterwards.

Per problem, medians of 9:  hot (back to back) | cold (640 MB fill, then the activation operand re-touched: what the
autotuner times and what the step sees) | cold + weights read by a separate kernel, then 48 MB of other traffic (pushes them
out of the L2s, not out of the MALL) | cold + weights read immediately before the launch (L2s of the reading XCDs + MALL)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops, packing

dev = "cuda"
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
junk = torch.empty(12 * 2 ** 20, dtype=torch.float32, device=dev)
ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)


def med(fn, prep, n=9):
    ts = []
    for _ in range(n):
        prep()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]


def touch(t):
    return t.view(torch.int32).sum()  # (one read of every line; the result is discarded)


PROBLEMS = [("unet 32^2 linear", 4096, 640, 640, 13, None), ("unet 64^2 linear", 16384, 320, 320, 9, None),
            ("unet 16^2 linear", 1024, 1280, 1280, 15, None), ("unet 64^2 ff1", 16384, 2560, 320, 9, None),
            ("unet 32^2 ff2", 4096, 640, 2560, 17, None), ("clip fc1", 4928, 3072, 768, 5, None), ("clip fc2", 4928, 768, 3072, 13, None),
            ("unet conv 64^2 320", 16384, 320, 2880, 18, (4, 64, 64, 320)), ("unet conv 32^2 640", 4096, 640, 5760, 18, (4, 32, 32, 640)),
            ("unet conv 16^2 1280", 1024, 1280, 11520, 18, (4, 16, 16, 1280))]
for name, M, N, K, tile, cv in PROBLEMS:
    if cv is None:
        A = torch.randn(M, K, device=dev, dtype=torch.float16)
        B = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.03
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        f = lambda: ops.gemm(A, B, out, tile_hint=tile, workspace=ws)
    else:
        Bn, H, W, Ci = cv
        A = torch.randn(Bn * H * W, Ci, device=dev, dtype=torch.float16)
        B = torch.randn(N, 9 * Ci, device=dev, dtype=torch.float16) * 0.02
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
        f = lambda: ops.gemm(A, B, out, M=M, conv=conv, tile_hint=tile, workspace=ws)
    f()
    f()
    hot = med(f, lambda: None)

    def p_cold():
        cold.fill_(0)
        A.add_(0)

    def p_mall():
        cold.fill_(0)
        touch(B)
        junk.add_(1)  # 48 MB read + 48 MB written: more than the eight L2s hold
        A.add_(0)

    def p_l2():
        cold.fill_(0)
        A.add_(0)
        touch(B)

    t_cold, t_mall, t_l2 = med(f, p_cold), med(f, p_mall), med(f, p_l2)
    print(f"{name:22s} M={M:6d} N={N:5d} K={K:6d} tile {tile:2d} weights {N * K * 2 / 2 ** 20:5.1f} MB: hot {hot:6.1f}  cold {t_cold:6.1f}  "
          f"weights in MALL {t_mall:6.1f}  weights just read {t_l2:6.1f} us", flush=True)
