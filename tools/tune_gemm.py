"""Dev tool: time every distinct GEMM/conv launch of the UNet schedule under each tile config."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from view_neti_amd import ops, sd_config as sc, synth
from view_neti_amd.engine.unet import UNetEngine

which = os.environ.get("ENGINE", "unet")
if which == "unet":
    cfg = sc.sd15().unet
    B, HW = 4, 64
    w = synth.unet_weights(cfg, device="cuda")
    eng = UNetEngine(cfg, w, B, HW, HW, autotune=False)
    del w
    eng.x_in.copy_(synth.gaussian((B, 4, HW, HW), 5))
    eng.timesteps.copy_(synth.timesteps(B))
    eng.ctx_k.copy_(synth.gaussian(tuple(eng.ctx_k.shape), 6).half())
    eng.ctx_v.copy_(synth.gaussian(tuple(eng.ctx_v.shape), 7).half())
    eng.dpred.copy_(synth.gaussian((B * HW * HW, 4), 8).half() * 0.01)
    eng.forward()
    eng.backward()
else:
    import bench, argparse
    args = argparse.Namespace(model="sd15", batch=4, resolution=512)
    import view_neti_amd.engine.schedule as S
    S.Schedule.autotune = lambda self, *a, **k: None
    _, step = bench.build_engine(args, 0, 1)
    step.step_eager()
    eng = step.vae if which == "vae" else step.text
torch.cuda.synchronize()

shapes = {}
for phase, lst in (("fwd", eng.fwd), ("bwd", eng.bwd)):
    for f in lst:
        fn = getattr(f, "func", None)
        if fn is not ops.gemm:
            continue
        kw = dict(f.keywords)
        A, Bm, out = f.args[:3]
        conv = kw.get("conv")
        M = kw.get("M") or A.shape[-2]
        N, K = Bm.shape[-2], Bm.shape[-1]
        key = (M, N, K, (conv["mode"], conv["stride"], conv["ups"]) if conv else None)
        shapes.setdefault(key, []).append(f)


def timeit(f, hint, reps=10):
    kw = dict(f.keywords)
    kw["tile_hint"] = hint
    for _ in range(2):
        ops.gemm(*f.args, **kw)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(*f.args, **kw)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3  # us


hints = [int(h) for h in os.environ.get("HINTS", "1,2,3,4").split(",")]
tot_best = tot_heur = 0.0
rows = []
for key, fs in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * len(kv[1])):
    M, N, K, conv = key
    f = fs[0]
    ts = {h: timeit(f, h) for h in hints}
    th = timeit(f, 0)
    best = min(ts, key=ts.get)
    gf = 2.0 * M * N * K / 1e9
    tot_best += ts[best] * len(fs)
    tot_heur += th * len(fs)
    rows.append((key, len(fs), gf, ts, th, best))
    print(f"M={M:6d} N={N:5d} K={K:6d} conv={str(conv):12s} x{len(fs):3d} {gf:7.2f}GF | " +
          " ".join(f"h{h}:{ts[h]:7.1f}us({gf / ts[h] * 1e3:5.0f}TF)" for h in ts) + f" | heur {th:7.1f} best h{best}")
print(f"total per step: heuristic {tot_heur / 1e3:.2f} ms, best-per-shape {tot_best / 1e3:.2f} ms")
os.makedirs("gpurun_out", exist_ok=True)
json.dump([(list(map(str, r[0])), r[1], r[2], r[3], r[4], r[5]) for r in rows], open("gpurun_out/tune_gemm.json", "w"))
