"""Dev tool: HBM-traffic counters for the train step's GEMM/conv launches.

rocprofv3 --pmc FETCH_SIZE on the whole bench process segfaults inside the profiler on this image (first
torch kernel of the weight generator), so the counter passes replay the step's GEMM problems standalone:

  python tools/pmc_gemm.py --dump  gpurun_out/gemm_specs.json      # builds the engine, records every GEMM launch
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/pmc_gemm.py --replay gpurun_out/gemm_specs.json

The replay allocates operands of the recorded shapes/strides filled with N(0,1) and issues the same
descriptors (same tile configuration, conv geometry, epilogue), one after the other, so each launch starts
with its operands evicted from L2 by its predecessors much as inside the step.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from view_neti_amd import ops

DT = {"torch.float16": torch.float16, "torch.float32": torch.float32, "torch.int32": torch.int32,
      "torch.int64": torch.int64}


def enc(v):
    if isinstance(v, torch.Tensor):
        return {"__t__": True, "shape": list(v.shape), "stride": list(v.stride()), "dtype": str(v.dtype)}
    if isinstance(v, dict):
        return {k: enc(x) for k, x in v.items()}
    return v


def dec(v, dev, extra=0):
    """`extra` elements of slack behind the view: batched launches address batch * stride past the first operand"""
    if isinstance(v, dict) and v.get("__t__"):
        n = 1 + sum((s - 1) * st for s, st in zip(v["shape"], v["stride"])) if v["shape"] else 1
        base = torch.empty(max(n, 1) + extra, dtype=DT[v["dtype"]], device=dev)
        if base.is_floating_point():
            base.normal_()
        else:
            base.zero_()
        return torch.as_strided(base, v["shape"], v["stride"])
    if isinstance(v, dict):
        return {k: dec(x, dev) for k, x in v.items()}
    return v


ap = argparse.ArgumentParser()
ap.add_argument("--dump")
ap.add_argument("--replay")
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
if a.dump:
    import bench
    args = argparse.Namespace(model="sd15", batch=4, resolution=512)
    _, eng = bench.build_engine(args, 0, 1)
    specs = []
    for f in eng.launches():
        if getattr(f, "func", None) is not ops.gemm:
            continue
        kw = {k: v for k, v in f.keywords.items() if k != "workspace"}
        specs.append({"args": [enc(x) for x in f.args], "kw": enc(kw)})
    json.dump(specs, open(a.dump, "w"))
    print(f"{len(specs)} GEMM launches -> {a.dump}")
else:
    dev = "cuda"
    ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)
    specs = json.load(open(a.replay))
    for _ in range(a.reps):
        for s in specs:
            kw = dec(s["kw"], dev)
            nb = max(int(kw.get("batch") or 1), 1) - 1
            extra = [nb * int(kw.get(k) or 0) for k in ("strideA", "strideB", "strideC")]
            args = [dec(x, dev, extra[i] if i < 3 else 0) for i, x in enumerate(s["args"])]
            ops.gemm(*args, workspace=ws, **kw)
    torch.cuda.synchronize()
    print(f"replayed {len(specs)} launches x {a.reps}")
