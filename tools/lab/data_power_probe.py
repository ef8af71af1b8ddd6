"""Dev tool (GPU box): is the dominant conv / GEMM tile POWER-bound or PIPELINE-bound?  The same launch is timed on random
operands (what the step computes on), on all-zero operands and on constant operands: identical instruction streams, so a
kernel that runs much faster on zeros is limited by the clock the 1400 W cap leaves (energy per FLOP), not by its schedule
(guide §5.4 rule 25: the 256^2 8-phase GEMM template gains 15-21 % on zeros).  Interleaved rounds, median of 7."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops, packing
dev = "cuda"
ws = torch.empty(64 * 2 ** 20, dtype=torch.float32, device=dev)


def t_us(fn, reps=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def fills(t, kind):
    if kind == "random":
        t.copy_(torch.randn(t.shape, device=dev) * (0.03 if t.dim() == 2 and t.shape[0] < 4096 else 1.0))
    elif kind == "zeros":
        t.zero_()
    else:
        t.fill_(0.5)


CONVS = [(4, 512, 512, 128, 128, 18), (4, 512, 512, 128, 128, 17), (4, 256, 256, 256, 256, 16), (4, 64, 64, 320, 320, 18),
         (4, 64, 64, 640, 640, 16)]
GEMMS = [(4096, 4096, 4096, 16), (16384, 320, 320, 9), (4096, 640, 640, 13), (4928, 3072, 768, 5), (16384, 2560, 320, 9)]
for (B, H, W, Ci, Co, tile) in CONVS:
    x = torch.empty(B * H * W, Ci, device=dev, dtype=torch.float16)
    w = torch.empty(Co, 9 * Ci, device=dev, dtype=torch.float16)
    y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    bias = torch.zeros(Co, device=dev)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
    f = lambda: ops.gemm(x, w, y, bias=bias, M=B * H * W, conv=conv, tile_hint=tile, workspace=ws, split_k=1)
    res = {k: [] for k in ("random", "zeros", "const")}
    for rnd in range(7):
        for kind in res:
            fills(x, kind)
            fills(w, kind)
            res[kind].append(t_us(f))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    gf = 2.0 * B * H * W * Co * 9 * Ci / 1e9
    print(f"conv {B}x{H}x{W} {Ci}->{Co} tile {tile}: " + "  ".join(f"{k} {v:7.1f}us {gf / v * 1e3:5.0f}TF" for k, v in med.items())
          + f"   zeros/random {med['random'] / med['zeros']:.2f}x", flush=True)
for (M, N, K, tile) in GEMMS:
    a = torch.empty(M, K, device=dev, dtype=torch.float16)
    b = torch.empty(N, K, device=dev, dtype=torch.float16)
    c = torch.empty(M, N, device=dev, dtype=torch.float16)
    f = lambda: ops.gemm(a, b, c, tile_hint=tile, workspace=ws, split_k=1)
    res = {k: [] for k in ("random", "zeros", "const")}
    for rnd in range(7):
        for kind in res:
            fills(a, kind)
            fills(b, kind)
            res[kind].append(t_us(f, reps=20))
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    gf = 2.0 * M * N * K / 1e9
    print(f"gemm {M}x{N}x{K} tile {tile}: " + "  ".join(f"{k} {v:7.1f}us {gf / v * 1e3:5.0f}TF" for k, v in med.items())
          + f"   zeros/random {med['random'] / med['zeros']:.2f}x", flush=True)
