"""Dev tool (GPU box): which launches of the train step do not scale with the batch.

    python tools/intercept_table.py <tag>        (writes gpurun_out/<tag>_intercept.json, prints the table)

The step at batch b costs about c + s*b (profiles/r04_batch_sweep.txt: 9.5 ms + 4.2 ms * b at 512^2) — a third of the
batch-4 step is the constant.  Both engines (batch 1 and batch 4) have the SAME launch list entry for entry (a split-K
GEMM and its reduce are one entry), so every entry is timed with HIP events at both sizes, in schedule order (operands
evicted by the predecessors, as in the step), and its constant part is estimated as (4*t1 - t4)/3.  Event pairs add the
same ~2 us to both sizes, which the difference t4 - t1 does not see.  (The list is queued behind 45 ms of filler work:
issued live, the host's ~25 us per launch is what the events would measure.)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from view_neti_amd import ops

tag = sys.argv[1] if len(sys.argv) > 1 else "r05x"
REPS = 5


def describe(f):
    fn, a, kw = getattr(f, "func", None), getattr(f, "args", ()), getattr(f, "keywords", {}) or {}
    name = getattr(fn, "__name__", None) or getattr(f, "__name__", None) or type(f).__name__
    if fn is ops.gemm:
        M, N, K, batch = bench.gemm_cost(f)[:4]
        conv = kw.get("conv")
        extra = []
        if conv:
            extra.append(f"conv{conv['Hi']}x{conv['Wi']}s{conv['stride']}m{conv['mode']}" + ("u" if conv.get("ups") else ""))
        for k in ("resid", "gate", "out2", "rowadd", "gn_sums"):
            if kw.get(k) is not None:
                extra.append(k)
        if kw.get("geglu"):
            extra.append(f"geglu{kw['geglu']}")
        return f"gemm {M}x{N}x{K}" + (f" b{batch}" if batch > 1 else "") + f" t{kw.get('tile_hint')} sk{kw.get('split_k')} " + ",".join(extra)
    shapes = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:2]
    ints = [x for x in a if isinstance(x, int)][:6]
    return f"{name} {shapes} {ints}"


def timed(eng):
    L = eng.launches()
    ts = [[] for _ in L]
    for _ in range(2):
        eng.step_eager()
    torch.cuda.synchronize()
    big = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    for _ in range(REPS):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in L]
        # the host needs ~25 us per (event, launch, event) triple, most kernels less: park the GPU behind ~45 ms of
        # matmuls so that the whole list is queued before its first launch starts and nothing waits for the host
        for _ in range(40):
            torch.mm(big, big)
        for (s, e), f in zip(ev, L):
            s.record()
            f()
            e.record()
        torch.cuda.synchronize()
        for i, (s, e) in enumerate(ev):
            ts[i].append(s.elapsed_time(e) * 1e3)
    return L, [sorted(t)[len(t) // 2] for t in ts]


res = {}
for b in (1, 4):
    args = argparse.Namespace(model="sd15", batch=b, resolution=512)
    _, eng = bench.build_engine(args, 0, 1)
    L, t = timed(eng)
    res[b] = ([describe(f) for f in L], t)
    del eng, L
    torch.cuda.empty_cache()
d1, t1 = res[1]
d4, t4 = res[4]
assert len(d1) == len(d4), (len(d1), len(d4))
rows = []
for i, (a, b, x, y) in enumerate(zip(d1, d4, t1, t4)):
    rows.append(dict(i=i, b1=a, b4=b, us1=x, us4=y, const_us=(4 * x - y) / 3, slope_us=(y - x) / 3))
tot1, tot4 = sum(t1), sum(t4)
summary = dict(sum_us_b1=tot1, sum_us_b4=tot4, const_ms=(4 * tot1 - tot4) / 3e3, slope_ms_per_sample=(tot4 - tot1) / 3e3,
               launches=len(rows))
# by kind
kinds = {}
for r in rows:
    k = r["b4"].split(" ")[0]
    if k == "gemm":
        M = int(r["b4"].split(" ")[1].split("x")[0])
        k = "gemm M>=65536" if M >= 65536 else f"gemm M={M}"
    g = kinds.setdefault(k, dict(n=0, us1=0.0, us4=0.0))
    g["n"] += 1
    g["us1"] += r["us1"]
    g["us4"] += r["us4"]
for g in kinds.values():
    g["const_us"] = (4 * g["us1"] - g["us4"]) / 3
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(summary=summary, kinds=kinds, rows=rows), open(f"gpurun_out/{tag}_intercept.json", "w"), indent=1)
print(json.dumps(summary))
print("\nby kind (sorted by constant part):")
for k, g in sorted(kinds.items(), key=lambda kv: -kv[1]["const_us"]):
    print(f"{g['const_us'] / 1e3:7.3f} ms const | b1 {g['us1'] / 1e3:7.3f} ms  b4 {g['us4'] / 1e3:7.3f} ms | {g['n']:4d} x {k}")
print("\ntop 60 launches by constant part:")
for r in sorted(rows, key=lambda r: -r["const_us"])[:60]:
    print(f"{r['const_us']:7.1f} us const | b1 {r['us1']:7.1f}  b4 {r['us4']:7.1f} | #{r['i']:4d} {r['b4'][:120]}")
