"""Dev tool: the halo-patch form of the 256x128 tile (tile_hint 18) against the row-major 8-phase tiles (17 / 16) on the
step's stride-1 3x3 convolutions, chunk-major K; checks bit-identity with tile 17."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops, packing
dev = "cuda"
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


CASES = [(2, 32, 48, 128, 128), (2, 32, 48, 192, 192), (4, 512, 512, 128, 128), (4, 256, 256, 256, 256), (4, 256, 256, 128, 256),
         (4, 128, 128, 512, 512), (4, 64, 64, 320, 320), (4, 64, 64, 640, 320), (4, 64, 64, 640, 640), (4, 32, 32, 640, 640)]
for (B, H, W, Ci, Co) in CASES:
    torch.manual_seed(0)
    x = torch.randn(B * H * W, Ci, device=dev).half()
    w4 = (torch.randn(Co, Ci, 3, 3) * 0.03).half()
    w = packing.conv3x3_fwd(w4, cm=True).to(dev)
    bias = torch.randn(Co, device=dev)
    y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
    mode = int(os.environ.get("MODE", "1"))  # 2: the stride-1 transposed gather (dgrad), same shapes
    if mode == 2:
        w = packing.conv3x3_dgrad(w4.permute(1, 0, 2, 3).contiguous(), cm=True).to(dev)
    conv = dict(mode=mode, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
    gf = 2.0 * B * H * W * Co * 9 * Ci / 1e9
    out, ref = [], None
    for h in (17, 18, 16):
        f = lambda: ops.gemm(x, w, y, bias=bias, M=B * H * W, conv=conv, tile_hint=h, workspace=ws, split_k=1)
        y.zero_()
        t = min(timeit(f) for _ in range(2))
        if ref is None:
            ref = y.clone()
            tag = ""
        else:
            tag = " (== 17)" if torch.equal(y, ref) else f" (rel {((y.float() - ref.float()).norm() / ref.float().norm()).item():.1e}, max |d| {(y.float() - ref.float()).abs().max().item():.1e})"
        out.append(f"h{h} {t:7.1f}us {gf / t * 1e3:5.0f}TF{tag}")
    print(f"conv {B}x{H}x{W} {Ci}->{Co} {gf:6.1f}GF: " + "  ".join(out), flush=True)
