"""Lab: how much of the VAE encoder (6.8 ms, few chip-filling launches) hides under the rest of the step (24.5 ms, ~1200
mostly under-filled launches) when the two run as concurrent graphs?  This is what cross-step pipelining — VAE(n+1) beside
text/UNet(n) — could buy: the VAE does not depend on the trainable state."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager(); torch.cuda.synchronize()

def capture(fn):
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    return g

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

g_full = capture(eng.step_eager)
vae_fwd = eng.vae.forward
g_vae = capture(vae_fwd)
eng.vae.forward = lambda: None          # the rest of the step reads the moments the last VAE run left behind
g_rest = capture(eng.step_eager)
eng.vae.forward = vae_fwd
t_full, t_vae, t_rest = timed(g_full.replay), timed(g_vae.replay), timed(g_rest.replay)
print(f"whole step {t_full:.2f} ms; VAE alone {t_vae:.2f}; rest alone {t_rest:.2f}; sum {t_vae + t_rest:.2f}", flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): g_rest.replay()
    with torch.cuda.stream(s2): g_vae.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)
t_both = timed(both)
print(f"VAE graph || rest graph: {t_both:.2f} ms per step  ({1e3 / t_both:.1f} steps/s vs {1e3 / t_full:.1f})", flush=True)
for prio in (0, -1):
    try:
        sp = torch.cuda.Stream(priority=prio); sq = torch.cuda.Stream(priority=-1 - prio)
        def both2():
            cur = torch.cuda.current_stream()
            sp.wait_stream(cur); sq.wait_stream(cur)
            with torch.cuda.stream(sp): g_rest.replay()
            with torch.cuda.stream(sq): g_vae.replay()
            cur.wait_stream(sp); cur.wait_stream(sq)
        print(f"  rest on priority {prio}, VAE on {-1 - prio}: {timed(both2):.2f} ms", flush=True)
    except Exception as e:
        print("  priority streams:", e)
