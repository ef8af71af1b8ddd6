"""Dev tool: implicit 3x3 convolutions of the step under the 256x256 tiles (5 generic, 16 8-phase) and the 256x128 rings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops
dev = "cuda"
HINTS = tuple(int(h) for h in os.environ.get("HINTS", "5,16,7,17").split(","))
ws = torch.empty(64 * 2 ** 20, dtype=torch.float32, device=dev)


def timeit(fn, reps=6):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (B, H, W, Ci, Co, mode) in [(4, 256, 256, 256, 256, 1), (4, 128, 128, 512, 512, 1), (4, 64, 64, 512, 512, 1), (4, 64, 64, 640, 640, 1),
                                (4, 64, 64, 640, 640, 2), (4, 64, 64, 960, 320, 1), (4, 32, 32, 1280, 1280, 1), (4, 512, 512, 128, 128, 1)]:
    x = torch.randn(B * H * W, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
    y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    ref = None
    conv = dict(mode=mode, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci)
    gf = 2.0 * B * H * W * Co * 9 * Ci / 1e9
    out = []
    for h in HINTS:
        t = min(timeit(lambda: ops.gemm(x, w, y, M=B * H * W, conv=conv, tile_hint=h, workspace=ws, split_k=1)) for _ in range(2))
        if ref is None:
            ref = y.float().clone()
            err = 0.0
        else:
            err = ((y.float() - ref).norm() / ref.norm()).item()
        out.append(f"h{h} {t:7.1f}us {gf / t * 1e3:5.0f}TF (d {err:.0e})")
    print(f"conv mode{mode} {H}x{W} {Ci}->{Co} {gf:6.1f}GF: " + " ".join(out), flush=True)
