import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
B, G = 4, 32
for HW, C in ((4096,320),(4096,640),(4096,960),(1024,320),(1024,640),(1024,960),(1024,1280),(1024,1920),(256,640),(256,1280),(256,1920),(256,2560),(64,1280),(64,2560)):
    x = torch.randn(B*HW, C, device="cuda").half(); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    mean = torch.zeros(B*G, device="cuda"); rstd = torch.zeros(B*G, device="cuda")
    ws = torch.zeros(ops.groupnorm_ws_floats(B, HW, C, G), device="cuda")
    res = []
    for small in (True, False):
        if small: os.environ.pop("VNETI_GN_NO_SMALL", None)
        else: os.environ["VNETI_GN_NO_SMALL"] = "1"
        def f():
            ops.groupnorm_fwd(x, y, gamma, beta, mean, rstd, ws, B, HW, C, G, 1e-5, True)
        def b():
            ops.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dx, ws, B, HW, C, G, True)
        for fn in (f, b):
            g = torch.cuda.CUDAGraph()
            fn(); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                for _ in range(10): fn()
            g.replay(); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / 10 * 1e3)
    print(f"HW {HW:5d} C {C:5d} slice {HW*C//G*2/1024:6.1f} KB | fwd one-launch {res[0]:6.1f} us vs 3-launch {res[2]:6.1f} | bwd {res[1]:6.1f} vs {res[3]:6.1f}")
