"""Dev tool: per-image cost of the input pipeline — host PIL path (compat/augment.py) vs the HIP kernels
(engine/input_pipeline.py) — on DTU-sized sources (1600x1200 -> 384x512, augmentation key 7) and 512^2 (key 5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from view_neti_amd.compat.augment import apply_plan, draw_plan
from view_neti_amd.engine.input_pipeline import DeviceImagePipeline

rng = np.random.default_rng(0)
for (sh, sw), size, key in (((1200, 1600), (384, 512), 7), ((1024, 1024), (512, 512), 5)):
    src = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
    pil = Image.fromarray(src)
    n = 40
    torch.manual_seed(0)
    plans = [draw_plan(key, size, size[1], size[0]) for _ in range(n)]
    t0 = time.time()
    for p in plans:
        im = pil.resize((size[1], size[0]), resample=Image.BICUBIC)
        im = apply_plan(im, p)
        arr = (np.array(im).astype(np.uint8) / 127.5 - 1.0).astype(np.float32)
        t = torch.from_numpy(arr).permute(2, 0, 1)
    host = (time.time() - t0) / n
    pipe = DeviceImagePipeline(*size)
    up = pipe.upload(src)
    out = torch.zeros(3, *size, device="cuda")
    for p in plans[:5]:
        pipe.run(up, out, resize=size, plan=p)
    torch.cuda.synchronize()
    t0 = time.time()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for p in plans:
        pipe.run(up, out, resize=size, plan=p)
    e.record(); torch.cuda.synchronize()
    wall = (time.time() - t0) / n
    print(f"{sh}x{sw} -> {size} key {key}: host PIL {host*1e3:.2f} ms/image | device {s.elapsed_time(e)/n*1e3:.0f} us GPU time, "
          f"{wall*1e3:.2f} ms wall/image (python launch-bound), avg {np.mean([len(p) for p in plans]):.1f} ops/plan")
