"""Dev tool (GPU box): do two parallel branches of ONE captured graph overlap on this runtime, and by how much, for the
launch shapes that make up the train step's latency floor?  Each case: a chain of `n` dependent launches of one GEMM shape
captured (a) as one branch of 2n launches, (b) as two parallel branches of n launches (fork / join through events)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops

dev = "cuda"
ws = [torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev) for _ in range(2)]


def chain(M, N, K, tile, n, wsp):
    a = [torch.randn(M, K, device=dev, dtype=torch.float16) for _ in range(2)]
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.03
    assert N == K

    def run():
        for i in range(n):
            ops.gemm(a[i & 1], w, a[(i + 1) & 1], tile_hint=tile, workspace=wsp, split_k=1)
    return run


def graph_of(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    return g


def t_ms(g, reps=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


side = torch.cuda.Stream()
for (M, N, K, tile) in [(4096, 640, 640, 13), (16384, 320, 320, 9), (1024, 1280, 1280, 15), (8192, 320, 320, 9), (2048, 640, 640, 13),
                        (16384, 1280, 1280, 16)]:
    n = 100
    c1, c2 = chain(M, N, K, tile, n, ws[0]), chain(M, N, K, tile, n, ws[1])

    def serial():
        c1()
        c2()

    def forked():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            c2()
        c1()
        main.wait_stream(side)

    ts, tf = t_ms(graph_of(serial)), t_ms(graph_of(forked))
    print(f"M={M} N={N} K={K} tile {tile}: 2x{n} launches serial {ts * 1e3 / (2 * n):6.2f} us/launch, two branches "
          f"{tf * 1e3 / (2 * n):6.2f} us/launch  ({ts / tf:.2f}x)", flush=True)
