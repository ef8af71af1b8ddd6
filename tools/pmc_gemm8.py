"""Dev tool: one plain GEMM (env M, N, K; default 4096^3) under tile HINT, launched REPS times on random f16 operands —
the body rocprofv3 --pmc passes wrap (tools/pmc_gemm8.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops
M, N, K = (int(os.environ.get(k, 4096)) for k in ("M", "N", "K"))
dev = "cuda"
A = torch.randn(M, K, device=dev).half()
B = torch.randn(N, K, device=dev).half()
C = torch.empty(M, N, device=dev, dtype=torch.float16)
for h in (int(x) for x in os.environ.get("HINTS", "5,16").split(",")):
    for _ in range(int(os.environ.get("REPS", 6))):
        ops.gemm(A, B, C, tile_hint=h, split_k=1)
    torch.cuda.synchronize()
