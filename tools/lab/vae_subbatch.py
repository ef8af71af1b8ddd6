"""Lab: would the VAE encoder be faster sample by sample?  At bs 4 the 512x512x128 activations are 268 MB each (more than the
256 MB MALL), at bs 1 67 MB.  Times one bs-4 forward against four bs-1 forwards (eager launches, all levels sub-batched,
so the 4x launch count of the small layers is charged too)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")
import torch
from view_neti_amd import sd_config as sc, synth
from view_neti_amd.engine.vae import VAEEncoderEngine

cfg = sc.CONFIGS["sd15"]()
w = synth.vae_weights(cfg.vae, device="cuda")


def timed(eng, reps, inner):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(inner): eng.forward()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(inner): eng.forward()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


e4 = VAEEncoderEngine(cfg.vae, w, 4, 512, 512)
e4.x_in.normal_()
t4 = timed(e4, 10, 1)
del e4
e2 = VAEEncoderEngine(cfg.vae, w, 2, 512, 512)
e2.x_in.normal_()
t2 = timed(e2, 10, 2)
del e2
e1 = VAEEncoderEngine(cfg.vae, w, 1, 512, 512)
e1.x_in.normal_()
t1 = timed(e1, 10, 4)
print(f"VAE encoder bs 4 in one pass: {t4:.3f} ms; two bs-2 passes: {t2:.3f} ms; four bs-1 passes: {t1:.3f} ms")
