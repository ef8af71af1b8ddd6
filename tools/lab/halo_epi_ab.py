"""Lab: tile 18 with the fused epilogues of the step (GroupNorm sums of the output, time-embedding row-add) beside the plain
one, on the VAE / UNet convolutions that use them.   [VNETI_LIB_PATH=...] python tools/lab/halo_epi_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)


def t_us(fn, reps=6):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (B, H, W, Ci, Co) in [(4, 512, 512, 128, 128), (4, 256, 256, 256, 256), (4, 128, 128, 512, 512), (4, 64, 64, 320, 320), (4, 64, 64, 640, 320)]:
    x = torch.randn(B * H * W, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
    y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    bias = torch.randn(Co, device=dev)
    radd = torch.randn(B, Co, device=dev).half()
    G, S = 32, 8
    sums = torch.zeros(B, S, G, 4, dtype=torch.int64, device=dev)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
    kw = dict(bias=bias, M=B * H * W, conv=conv, tile_hint=18, workspace=ws, split_k=1)
    gn = dict(gn_sums=sums, gn_hw=H * W, gn_groups=G, gn_slots=S)
    ra = dict(rowadd=radd, rows_per_group=H * W)
    res = []
    sums64 = torch.zeros(B, 64, G, 4, dtype=torch.int64, device=dev)
    gn64 = dict(gn_sums=sums64, gn_hw=H * W, gn_groups=G, gn_slots=64)
    for name, extra in (("plain", {}), ("rowadd(EPI1)", ra), ("gn", gn), ("gn slots64", gn64), ("gn+rowadd", dict(gn, **ra))):
        f = lambda: ops.gemm(x, w, y, **kw, **extra)
        res.append(f"{name} {min(t_us(f) for _ in range(3)):7.1f}us")
    print(f"conv {B}x{H}x{W} {Ci}->{Co}: " + "  ".join(res), flush=True)
