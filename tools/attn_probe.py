"""Dev tool: run the UNet's 64x64 self-attention (B=4,H=8,N=4096,D=40) fwd/bwd kernels a few times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops
B, H, N, D = int(os.environ.get("B", 4)), int(os.environ.get("H", 8)), int(os.environ.get("N", 4096)), int(os.environ.get("D", 40))
Nk = int(os.environ.get("NK", N))
C = H * D
dev = "cuda"
ops.set_default_gemm_workspace(torch.empty(int(os.environ.get('WS_M', 16)) * 2**20, dtype=torch.float32, device=dev))
q = torch.randn(B * N, C, device=dev).half(); k = torch.randn(B * Nk, C, device=dev).half(); v = torch.randn(B * Nk, C, device=dev).half()
do = torch.randn(B * N, C, device=dev).half()
o = torch.zeros_like(q); lse = torch.zeros(B, H, N, device=dev); delta = torch.zeros(B, H, N, device=dev)
dq = torch.zeros_like(q); dk = torch.zeros_like(k); dv = torch.zeros_like(k)
sc = D ** -0.5
def run():
    ops.attn_fwd(q, k, v, o, lse, B, H, N, Nk, D, sc, False)
    ops.attn_bwd_dq(q, k, v, do, lse, delta, dq, B, H, N, Nk, D, sc, False, O=o)
    ops.attn_bwd_dkv(q, k, v, do, lse, delta, dk, dv, B, H, N, Nk, D, sc, False)
for _ in range(3): run()
torch.cuda.synchronize()
import time
for name, fn in (("fwd", lambda: ops.attn_fwd(q, k, v, o, lse, B, H, N, Nk, D, sc, False)),
                 ("dq", lambda: ops.attn_bwd_dq(q, k, v, do, lse, delta, dq, B, H, N, Nk, D, sc, False, O=o)),
                 ("dkv", lambda: ops.attn_bwd_dkv(q, k, v, do, lse, delta, dk, dv, B, H, N, Nk, D, sc, False))):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    fl = 4.0 * B * H * N * Nk * D * (1 if name == "fwd" else (2 if name == "dq" else 2.5)) / 1e9  # real (unpadded) GF
    print(f"attn {name:4s} N={N} Nk={Nk} D={D}: {t*1e3:8.1f} us  {fl/t:7.1f} TF/s (unpadded flops)")
