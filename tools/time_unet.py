"""Dev tool: time the SD-1.5 UNet engine forward/backward on one GPU (HIP events)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import sd_config as sc, synth
from view_neti_amd.engine.unet import UNetEngine

name = sys.argv[1] if len(sys.argv) > 1 else "sd15"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
HW = int(sys.argv[3]) if len(sys.argv) > 3 else 64
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
cfg = sc.CONFIGS[name]().unet
t0 = time.time()
w = synth.unet_weights(cfg, device="cuda")
torch.cuda.synchronize()
print(f"weights {time.time()-t0:.1f}s")
t0 = time.time()
eng = UNetEngine(cfg, w, B, HW, HW)
del w
torch.cuda.synchronize()
print(f"engine build {time.time()-t0:.1f}s, {eng.bytes/2**30:.2f} GiB, {len(eng.fwd)} fwd / {len(eng.bwd)} bwd launches")
eng.x_in.copy_(synth.gaussian((B, 4, HW, HW), 5))
eng.timesteps.copy_(synth.timesteps(B))
eng.ctx_k.copy_(synth.gaussian(tuple(eng.ctx_k.shape), 6).half())
eng.ctx_v.copy_(synth.gaussian(tuple(eng.ctx_v.shape), 7).half())
eng.dpred.copy_(synth.gaussian((B * HW * HW, 4), 8).half() * 0.01)
for _ in range(2):
    eng.forward(); eng.backward()
torch.cuda.synchronize()
print("pred finite:", torch.isfinite(eng.pred.float()).all().item(), "std", eng.pred.float().std().item(),
      "dctx finite:", torch.isfinite(eng.dctx_k.float()).all().item(), eng.dctx_k.float().abs().mean().item())
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(iters):
    e[0].record(); eng.forward(); e[1].record(); eng.backward(); e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
print(f"eager: fwd {tf/iters:.2f} ms  bwd {tb/iters:.2f} ms")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng.forward(); eng.backward()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        eng.forward(); eng.backward()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(iters):
    g.replay()
torch.cuda.synchronize()
print(f"graph: fwd+bwd {(time.time()-t0)/iters*1e3:.2f} ms")
