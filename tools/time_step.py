"""Dev tool: per-phase time of the train step (each phase replayed from its own hipGraph)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="sd15"); ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--resolution", type=int, default=512); ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
cfg, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()

def timed(name, fn):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.iters): g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    print(f"{name:18s} {ms:8.3f} ms")
    return ms

tot = 0
tot += timed("vae.forward", eng.vae.forward)
tot += timed("text.forward", eng.text.forward)
tot += timed("unet.forward", eng.unet.forward)
tot += timed("unet.backward", eng.unet.backward)
tot += timed("text.backward", eng.text.backward)
tot += timed("optimizer", eng.optimizer_step)
print(f"{'sum':18s} {tot:8.3f} ms")
timed("whole step", eng.step_eager)
eng.overlap = False
timed("whole (1 stream)", eng.step_eager)
eng.overlap = True
# kernel-class breakdown inside the UNet
from view_neti_amd import ops
def cls(f):
    fn = getattr(f, "func", None)
    return getattr(fn, "__name__", "lambda")
for nm, lst in (("vae.fwd", eng.vae.fwd), ("text.fwd", eng.text.fwd), ("unet.pre", eng.unet.fwd_pre), ("unet.fwd", eng.unet.fwd), ("unet.bwd", eng.unet.bwd), ("text.bwd", eng.text.bwd)):
    groups = {}
    for f in lst: groups.setdefault(cls(f), []).append(f)
    line = f"{nm:9s}"
    for k, fs in sorted(groups.items()):
        for f in fs: f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(3):
            for f in fs: f()
        e.record(); torch.cuda.synchronize()
        line += f" {k}:{s.elapsed_time(e)/3:.2f}ms/{len(fs)}"
    print(line)
