"""Dev tool (GPU box): the row-stationary linear kernel (tile_hint 19) against every tiled kernel on the short-K linears of
the step, cold (a 640 MB fill between timed launches, as the autotuner does) — and the fused LayerNorm launch against
LayerNorm + GEMM.   python tools/linear_ab.py [tag]"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from view_neti_amd import ops, packing

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
DEV = "cuda"
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=DEV)


def timed(fn, touch=None, reps=9):
    ts = []
    for _ in range(reps):
        cold.fill_(0)
        if touch is not None:
            touch.add_(0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[len(ts) // 2]


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).half()


rows = []
CASES = [("to_out 64^2", 16384, 320, 320, "resid"), ("to_out 32^2", 4096, 640, 640, "resid"), ("qkv 64^2", 16384, 960, 320, ""),
         ("qkv 32^2", 4096, 1920, 640, ""), ("ff1 64^2", 16384, 2560, 320, "geglu1"), ("ff1 32^2", 4096, 5120, 640, "geglu1"),
         ("ff2-dgrad 64^2", 16384, 1280, 320, "geglu2"), ("ff2-dgrad 32^2", 4096, 2560, 640, "geglu2"),
         ("clip qkv", 4928, 2304, 768, ""), ("clip fc1", 4928, 3072, 768, "out2"), ("clip out", 4928, 768, 768, "resid"),
         ("clip fc2-dgrad", 4928, 3072, 768, "gate")]
for name, M, N, K, epi in CASES:
    A, B = rnd(M, K), rnd(N, K, scale=1 / math.sqrt(K))
    bias = torch.randn(N, device=DEV)
    kw = dict(bias=bias)
    if epi == "resid":
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        kw["resid"] = rnd(M, N)
    elif epi == "geglu1":
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        kw.update(out2=torch.zeros(M, N // 2, dtype=torch.float16, device=DEV), geglu=1)
    elif epi == "geglu2":
        out = torch.zeros(M, 2 * N, dtype=torch.float16, device=DEV)
        kw = dict(gate=rnd(M, 2 * N), gate_act=ops.ACT_GELU, geglu=2)
    elif epi == "out2":
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        kw.update(out2=torch.zeros(M, N, dtype=torch.float16, device=DEV), act2=ops.ACT_QUICK_GELU)
    elif epi == "gate":
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        kw = dict(gate=rnd(M, N), gate_act=ops.ACT_QUICK_GELU)
    else:
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        kw = {}
    best = (None, 1e9)
    for h in (1, 2, 3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 16, 17):
        t = timed(lambda: ops.gemm(A, B, out, tile_hint=h, split_k=1, **kw), touch=A)
        if t < best[1]:
            best = (h, t)
    t19 = timed(lambda: ops.gemm(A, B, out, tile_hint=19, **kw), touch=A)
    row = dict(case=name, M=M, N=N, K=K, epi=epi, best_tiled=best[0], best_tiled_us=best[1], tile19_us=t19)
    if epi in ("", "geglu1"):  # the launches a LayerNorm feeds
        g, b = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
        y = torch.zeros(M, K, dtype=torch.float16, device=DEV)
        mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        row["ln_plus_best_tiled_us"] = timed(lambda: (ops.layernorm_fwd(A, y, g, b, mean, rstd, 1e-5),
                                                     ops.gemm(y, B, out, tile_hint=best[0], split_k=1, **kw)), touch=A)
        row["fused_ln_tile19_us"] = timed(lambda: ops.gemm(A, B, out, tile_hint=19, ln=(g, b, mean, rstd, 1e-5), **kw), touch=A)
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(f"gpurun_out/{tag}_linear_ab.json", "w"), indent=1)
