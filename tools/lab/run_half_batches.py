"""Lab: does running the step as TWO concurrent half-batch graphs (2 x bs=2 on two streams) beat ONE bs=4 graph?
The samples of a batch are independent through VAE / CLIP / UNet (GroupNorm is per sample), so the split is exact;
the hope is that the two launch chains desynchronise each other's prologue/epilogue memory phases."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

full = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, e4 = bench.build_engine(full, 0, 1)
e4.capture()
t4 = timed(e4.step)
print(f"one graph, bs=4: {t4:.2f} ms/step", flush=True)
del e4
torch.cuda.empty_cache()
half = argparse.Namespace(model="sd15", batch=2, resolution=512)
_, a = bench.build_engine(half, 0, 1)
_, b = bench.build_engine(half, 1, 1)
a.capture(); b.capture()
ta = timed(a.step)
print(f"one graph, bs=2: {ta:.2f} ms/step (x2 sequential = {2*ta:.2f})", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): a.step()
    with torch.cuda.stream(s2): b.step()
    cur.wait_stream(s1); cur.wait_stream(s2)
tb = timed(both)
print(f"two concurrent graphs, 2 x bs=2: {tb:.2f} ms per pair  (vs {t4:.2f} for bs=4)", flush=True)
