"""Dev tool (GPU box): how much of the train step is per-launch latency that a second stream can hide?
The step costs ~9.5 ms + 4.2 ms x batch (bench.py --batch 1/2/4/8): the constant is the sum of the ramp / tail / latency
chains of ~1100 launches, most of which underfill the chip.  Probe: two independent batch-2 steps replayed concurrently on
two streams against one batch-4 step on one stream — the same images per unit of time if the streams overlap perfectly."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=60)
a = ap.parse_args()


def mk(batch):
    args = argparse.Namespace(model="sd15", batch=batch, resolution=512)
    _, eng = bench.build_engine(args, 0, 1)
    eng.capture()
    return eng


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


e4 = mk(4)
t4 = timed(e4.step, a.iters)
print(f"one batch-4 step, one stream: {t4:.2f} ms", flush=True)
del e4
torch.cuda.empty_cache()
ea, eb = mk(2), mk(2)
t2 = timed(ea.step, a.iters)
print(f"one batch-2 step, one stream: {t2:.2f} ms", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        ea.step()
    with torch.cuda.stream(s2):
        eb.step()


tb = timed(both, a.iters)
print(f"two batch-2 steps on two streams: {tb:.2f} ms per pair  ({t4 / tb:.2f}x the batch-4 step's image rate)", flush=True)
