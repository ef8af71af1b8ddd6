"""Dev tool: one implicit conv of the step timed COLD (a 640 MB fill between launches, as the autotuner does) under every
tile, with the GroupNorm-sums epilogue the VAE launches carry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.empty(64 * 2 ** 20, dtype=torch.float32, device=dev)
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
for (B, H, W, Ci, Co) in [(4, 512, 512, 128, 128), (4, 256, 256, 256, 256), (4, 256, 256, 128, 256), (4, 128, 128, 512, 512)]:
    x = torch.randn(B * H * W, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
    bias = torch.randn(Co, device=dev)
    y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    sums = torch.zeros(B, 8, 32, 4, dtype=torch.int64, device=dev)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci)
    gf = 2.0 * B * H * W * Co * 9 * Ci / 1e9
    out = []
    from view_neti_amd import packing
    w4 = (torch.randn(Co, Ci, 3, 3) * 0.03).half()
    for h, ko in ((7, 0), (16, 0), (17, 0), (16, 1), (17, 1), (18, 1)):
        w = packing.conv3x3_fwd(w4, cm=bool(ko)).to(dev)
        conv["korder"] = ko
        ts = []
        for _ in range(7):
            cold.fill_(0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.gemm(x, w, y, M=B * H * W, conv=conv, tile_hint=h, workspace=ws, split_k=1, bias=bias, gn_sums=sums, gn_hw=H * W,
                     gn_groups=32, gn_slots=8)
            e.record(); e.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        t = sorted(ts)[3]
        out.append(f"h{h}{'cm' if ko else ''} {t:6.1f}us {gf / t * 1e3:4.0f}TF")
    print(f"conv {H}x{W} {Ci}->{Co} {gf:6.1f}GF cold+gn: " + " ".join(out), flush=True)
