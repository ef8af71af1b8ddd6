"""Lab (round 6): does a CU-masked side stream let the NEXT batch's VAE encode hide beside this step?

Round 2 replayed the VAE graph and the graph of the rest of the step on two plain streams and gained nothing: the encoder's
one-block-per-CU tiles take every CU and the chain's short launches queue behind them.  Here the VAE graph runs on a stream
created with hipExtStreamCreateWithCUMask, so it can only ever hold `k` CUs of every XCD (or n whole XCDs).

Prints, same box, interleaved:
  whole step / VAE alone / rest alone (one stream each)
  VAE alone on each masked stream            -> does the mask bind a graph launch?  (expect ~256/n_cus slower)
  pair = rest (unmasked)  ||  VAE (masked)   -> ms per pair; the pipelined step would cost this
  pair = rest (complement) || VAE (masked)   -> the same with the main chain confined to the other CUs
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from view_neti_amd import streams as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()

os.environ["VNETI_OVERLAP"] = "0"  # linear graphs only (the product default since round 6): a graph with a fork takes runtime-internal (unmasked) streams
args = argparse.Namespace(model="sd15", batch=4, resolution=512)
_, eng = bench.build_engine(args, 0, 1)
eng.step_eager()
torch.cuda.synchronize()


def capture(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    return g


def timed(fn, iters=a.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


g_full = capture(eng.step_eager)
vae_fwd = eng.vae.forward
g_vae = capture(vae_fwd)
eng.vae.forward = lambda: None  # the rest of the step reads the moments the last VAE run left behind
g_rest = capture(eng.step_eager)
eng.vae.forward = vae_fwd

plain = torch.cuda.Stream()


def on(stream, g):
    def f():
        with torch.cuda.stream(stream):
            g.replay()
    return f


def pair(s_rest, s_vae):
    def f():
        cur = torch.cuda.current_stream()
        s_rest.wait_stream(cur)
        s_vae.wait_stream(cur)
        with torch.cuda.stream(s_rest):
            g_rest.replay()
        with torch.cuda.stream(s_vae):
            g_vae.replay()
        cur.wait_stream(s_rest)
        cur.wait_stream(s_vae)
    return f


full_w = [(1 << 32) - 1] * 8
variants = []
for k in (8, 12, 16, 20, 24):
    m = S.per_xcd_mask(k)
    variants.append((f"per-XCD {k:2d} ({8 * k:3d} CUs)", m, [full_w[i] & ~m[i] for i in range(8)]))
for n in (2, 3, 4):
    m = S.whole_xcd_mask(n)
    variants.append((f"whole XCDs {n} ({32 * n:3d} CUs)", m, [full_w[i] & ~m[i] for i in range(8)]))

for rnd in range(a.rounds):
    t_full, t_vae, t_rest = timed(g_full.replay), timed(g_vae.replay), timed(g_rest.replay)
    print(f"[round {rnd}] whole step {t_full:.2f} ms; VAE alone {t_vae:.2f}; rest alone {t_rest:.2f}; sum {t_vae + t_rest:.2f}",
          flush=True)
    t_plain = timed(pair(plain, torch.cuda.Stream()))
    print(f"  rest || VAE on two PLAIN streams: {t_plain:.2f} ms per pair", flush=True)
    for name, m, comp in variants:
        sv = S.CUMaskStream(m)
        assert sv.runtime_mask() == m, (sv.runtime_mask(), m)
        sc_ = S.CUMaskStream(comp)
        t_v = timed(on(sv.stream, g_vae))
        t_r = timed(on(sc_.stream, g_rest))
        t_p = timed(pair(plain, sv.stream))
        t_pc = timed(pair(sc_.stream, sv.stream))
        print(f"  {name}: VAE masked alone {t_v:6.2f} | rest on complement alone {t_r:6.2f} | rest(unmasked)||VAE(masked) "
              f"{t_p:6.2f} ({1e3 / t_p:.1f} steps/s) | rest(complement)||VAE(masked) {t_pc:6.2f} ({1e3 / t_pc:.1f}) "
              f"| sequential {t_full:.2f} ({1e3 / t_full:.1f})", flush=True)
        sv.close()
        sc_.close()
