"""Lab: the CLIP text encoder's GEMMs (M = 16 layers x 4 samples x 77 tokens = 4928) under the candidate tiles, cache-hot and
cold (640 MB fill between launches), with their real epilogues: fc1 = bias + quick-GELU second output (EPI 2), out-proj / fc2 =
f32 output + f32 residual, qkv = bias."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
M = 4928
for name, N, K, kind in (("fc1", 3072, 768, "act2"), ("qkv", 2304, 768, "plain"), ("fc2", 768, 3072, "f32res"), ("out", 768, 768, "f32res"),
                         ("dfc1", 768, 3072, "plain"), ("dfc2", 3072, 768, "gate")):
    a = torch.randn(M, K, device=dev).half()
    b = (torch.randn(N, K, device=dev) * 0.03).half()
    bias = torch.randn(N, device=dev)
    out = []
    for h in (5, 16, 17, 13, 9, 7, 1):
        if kind == "f32res":
            c = torch.empty(M, N, device=dev, dtype=torch.float32); r = torch.randn(M, N, device=dev)
            f = lambda: ops.gemm(a, b, c, bias=bias, resid=r, tile_hint=h, workspace=ws, split_k=1)
        elif kind == "act2":
            c = torch.empty(M, N, device=dev, dtype=torch.float16); c2 = torch.empty(M, N, device=dev, dtype=torch.float16)
            f = lambda: ops.gemm(a, b, c, bias=bias, out2=c2, act2=2, tile_hint=h, workspace=ws, split_k=1)
        elif kind == "gate":
            c = torch.empty(M, N, device=dev, dtype=torch.float16); gsrc = torch.randn(M, N, device=dev).half()
            f = lambda: ops.gemm(a, b, c, gate=gsrc, gate_act=2, tile_hint=h, workspace=ws, split_k=1)
        else:
            c = torch.empty(M, N, device=dev, dtype=torch.float16)
            f = lambda: ops.gemm(a, b, c, bias=bias, tile_hint=h, workspace=ws, split_k=1)
        try:
            f(); f()
        except RuntimeError as e:
            out.append(f"h{h} n/a"); continue
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            f()
        e.record(); torch.cuda.synchronize()
        hot = s.elapsed_time(e) / 20 * 1e3
        ts = []
        for _ in range(7):
            cold.fill_(0); a.add_(0)
            s.record(); f(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3)
        out.append(f"h{h} {hot:5.1f}/{sorted(ts)[3]:5.1f}")
    print(f"{name:5s} {M}x{N}x{K} {kind:7s}: hot/cold us  " + "  ".join(out), flush=True)
