"""Lab: an attention-forward variant behind an environment switch (argv[1], e.g. VNETI_ATTN_WIDE with
tools/lab/attic/attn_fwd_wide.patch applied) against the product forward on the step's long self-attentions: bit-equality of O
and lse, then interleaved timing.   python tools/lab/attn_env_ab.py VNETI_ATTN_WIDE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops

dev = "cuda"
ENV = sys.argv[1] if len(sys.argv) > 1 else "VNETI_ATTN_WIDE"
for (B, H, N, Nk, D) in [(4, 8, 4096, 4096, 40), (1, 8, 4096, 4096, 40), (4, 8, 4096, 77, 40), (2, 5, 4096, 4096, 64), (1, 5, 9216, 9216, 64), (2, 8, 2048, 1000, 40)]:
    C = H * D
    g = torch.Generator().manual_seed(N + D)
    q = torch.randn(B * N, C, generator=g).half().to(dev); k = torch.randn(B * Nk, C, generator=g).half().to(dev)
    v = torch.randn(B * Nk, C, generator=g).half().to(dev)
    sc = D ** -0.5
    outs = {}
    for wide in ("0", "1"):
        os.environ[ENV] = wide
        o = torch.full_like(q, 7.0); lse = torch.zeros(B, H, N, device=dev)
        ops.attn_fwd(q, k, v, o, lse, B, H, N, Nk, D, sc, False)
        torch.cuda.synchronize()
        outs[wide] = (o, lse)
    same = torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])
    nbad = (outs["0"][0] != outs["1"][0]).sum().item()
    ts = {"0": [], "1": []}
    for rnd in range(3):
        for wide in ("0", "1"):
            os.environ[ENV] = wide
            o, lse = outs[wide]
            for _ in range(2): ops.attn_fwd(q, k, v, o, lse, B, H, N, Nk, D, sc, False)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): ops.attn_fwd(q, k, v, o, lse, B, H, N, Nk, D, sc, False)
            e.record(); torch.cuda.synchronize()
            ts[wide].append(s.elapsed_time(e) * 100)
    print(f"B{B} H{H} N{N} Nk{Nk} D{D}: bit-equal {same} (diff elems {nbad}, finite {bool(torch.isfinite(outs['1'][0].float()).all())}) | product "
          f"{min(ts['0']):.1f}-{max(ts['0']):.1f} us  wide {min(ts['1']):.1f}-{max(ts['1']):.1f} us", flush=True)
