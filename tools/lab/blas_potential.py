"""Lab: for every plain (non-conv, unbatched) GEMM launch of the train step, the pinned launch of this repo's kernel against
torch.matmul (hipBLASLt / rocBLAS) on the same operands, both timed COLD (640 MB fill between launches, median of 5):
how much of the step a library back-end for plain GEMMs could buy, fused epilogues not counted against the library."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")
import torch
import bench
from view_neti_amd import ops

args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device="cuda")


def t_cold(fn, reps=5):
    ts = []
    for _ in range(reps):
        cold.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


agg = collections.OrderedDict()
for phase, lst in (("vae", eng.vae.fwd), ("text.f", eng.text.fwd), ("unet.p", eng.unet.fwd_pre), ("unet.f", eng.unet.fwd),
                   ("unet.b", eng.unet.bwd), ("text.b", eng.text.bwd)):
    for f in lst:
        if getattr(f, "func", None) is not ops.gemm:
            continue
        kw = f.keywords
        if kw.get("conv") or (kw.get("batch") or 0) > 1:
            continue
        A, Bm, out = f.args[:3]
        if A.dtype != torch.float16 or A.dim() != 2 or Bm.dim() != 2:
            continue
        M, K = A.shape
        N = Bm.shape[0]
        fused = [k for k in ("gate", "out2", "gn_sums", "rowadd") if kw.get(k) is not None] + (["geglu"] if kw.get("geglu") else [])
        key = (phase, M, N, K, out.dtype == torch.float32, bool(kw.get("resid") is not None), tuple(fused))
        if key in agg:
            agg[key][0] += 1
            continue
        ours = t_cold(f)
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
        Bt = Bm.t()
        blas = t_cold(lambda: torch.matmul(A, Bt, out=C))
        agg[key] = [1, ours, blas]
tot_o = tot_b = gain_plain = gain_all = 0.0
for k, (n, o, b) in sorted(agg.items(), key=lambda kv: -kv[1][0] * max(kv[1][1] - kv[1][2], 0)):
    ph, M, N, K, f32, res, fused = k
    tot_o += n * o; tot_b += n * min(o, b)
    g = n * max(o - b, 0)
    gain_all += g
    if not fused:
        gain_plain += g
    print(f"{ph:7s} M={M:6d} N={N:5d} K={K:6d} f32out={int(f32)} resid={int(res)} fused={','.join(fused) or '-':12s} x{n:3d} "
          f"ours {o:7.1f}us blas {b:7.1f}us gain {g/1e3:6.3f}ms")
print(f"plain-GEMM launches: ours {tot_o/1e3:.2f} ms; best-of {tot_b/1e3:.2f} ms; gain if library where faster: "
      f"{gain_all/1e3:.2f} ms (only launches without fused epilogue extras: {gain_plain/1e3:.2f} ms)")
