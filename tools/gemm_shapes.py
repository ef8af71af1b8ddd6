"""Dev tool: per-problem time of every GEMM/conv launch of the train step under its pinned configuration."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from view_neti_amd import ops

args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()
agg = collections.OrderedDict()
for phase, lst in (("vae", eng.vae.fwd), ("text.f", eng.text.fwd), ("unet.p", eng.unet.fwd_pre), ("unet.f", eng.unet.fwd),
                   ("unet.b", eng.unet.bwd), ("text.b", eng.text.bwd)):
    for f in lst:
        if getattr(f, "func", None) is not ops.gemm:
            continue
        kw = f.keywords
        A, Bm = f.args[0], f.args[1]
        M = kw.get("M") or A.shape[-2]
        N, K = kw.get("N") or Bm.shape[-2], kw.get("K") or Bm.shape[-1]
        conv = kw.get("conv")
        key = (phase, M, N, K, kw.get("batch") or 1, (conv["mode"], conv["stride"], conv["ups"]) if conv else None,
               kw.get("tile_hint"), kw.get("split_k"))
        for _ in range(2): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 5 * 1e3
        a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += t
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print(f"total GEMM time (isolated launches) {tot/1e3:.2f} ms over {sum(v[0] for v in agg.values())} launches")
for k, (n, t) in rows[:45]:
    ph, M, N, K, b, conv, tile, sk = k
    gf = 2.0 * M * N * K * b / 1e9
    print(f"{ph:7s} M={M:7d} N={N:5d} K={K:6d} b={b:2d} conv={str(conv):10s} tile={tile} sk={sk} x{n:3d} {t/n:7.1f}us each {gf/(t/n)*1e3:5.0f}TF tot {t/1e3:5.2f}ms")
