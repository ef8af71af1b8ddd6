"""Lab (round 6, candidate (c) of round 4's table): the UPPER BOUND of "the next launch's weights requested early" over the
whole train step.  Every GEMM launch of the step's schedule (deduplicated by problem) is timed cold — 640 MB fill, activation
operand re-touched, as the autotuner does and as the step sees it: 3.4 GB of packed weights stream through a 256 MB MALL every
step — and with its weights read by another kernel beforehand and pushed out of the L2s but not out of the MALL.  The sum of
(cold - MALL-resident) x launches is what a perfect, free prefetch of every weight matrix could return."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from view_neti_amd import ops
from view_neti_amd.roofline import gemm_cost

args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager()
torch.cuda.synchronize()
dev = "cuda"
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
junk = torch.empty(12 * 2 ** 20, dtype=torch.float32, device=dev)


def med(fn, prep, n=7):
    ts = []
    for _ in range(n):
        prep()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]


groups = {}
for f in eng.launches():
    if getattr(f, "func", None) is not ops.gemm:
        continue
    M, N, K, batch, flops, nbytes, _, _ = gemm_cost(f)
    kw = f.keywords
    key = (M, N, K, batch, bool(kw.get("conv")), kw.get("tile_hint"), kw.get("split_k"))
    groups.setdefault(key, []).append(f)
rows = []
for key, fs in groups.items():
    f = fs[0]
    A, B = f.args[0], f.args[1]

    def p_cold():
        cold.fill_(0)
        A.add_(0)

    def p_mall():
        cold.fill_(0)
        B.float().sum()  # one read of every weight line
        junk.add_(1)
        A.add_(0)

    f()
    t_c, t_m = med(f, p_cold), med(f, p_mall)
    rows.append((key, len(fs), t_c, t_m, B.numel() * 2 / 2 ** 20))
tot_c = sum(n * c for _, n, c, _, _ in rows)
tot_gain = sum(n * max(c - m, 0.0) for _, n, c, m, _ in rows)
net = sum(n * (c - m) for _, n, c, m, _ in rows)
print(f"{len(rows)} distinct GEMM problems, {sum(n for _, n, *_ in rows)} launches; cold sum {tot_c / 1e3:.2f} ms per step")
print(f"sum of (cold - weights-in-MALL) over the launches that gain: {tot_gain / 1e3:.3f} ms; signed sum {net / 1e3:.3f} ms")
print("largest:")
for key, n, c, m, mb in sorted(rows, key=lambda r: -r[1] * (r[2] - r[3]))[:25]:
    print(f"  M={key[0]:6d} N={key[1]:5d} K={key[2]:6d} conv={int(key[4])} tile {key[5]} split {key[6]}  x{n:3d}  weights {mb:6.1f} MB  "
          f"cold {c:7.1f}  MALL {m:7.1f}  -> {n * (c - m):7.1f} us/step")
