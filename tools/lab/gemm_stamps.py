"""[needs the lab switches: git apply tools/lab/attic/lab_switches.patch first — tools/lab/README.md]
Lab: where a block of the generic GEMM tiles spends its life on the step's short-K linears, COLD (weights evicted, the
activation operand re-touched: as inside the step).  Builds csrc/gemm_conv.hip with -DVN_GEMM_STAMP into
tools/lab/libvneti_gstamp.so (thread 0 of every block records s_memtime at: entry, prologue DMAs issued, first stage landed,
loop end, C tile in LDS, stores issued, stores acknowledged; 4-stage ring tiles 13 / 14 / 15 only) and prints per-section medians.
    python tools/lab/gemm_stamps.py build          (in the container)
    VNETI_LIB_PATH=tools/lab/libvneti_gstamp.so python tools/lab/gemm_stamps.py   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "view_neti_amd", "csrc")
SO = os.path.join(ROOT, "tools", "lab", "libvneti_gstamp.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    obj = "/tmp/gemm_conv_stamp.o"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DVN_GEMM_STAMP", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-c",
                           os.path.join(CS, "gemm_conv.hip"), "-o", obj])
    objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != "gemm_conv.o"]
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", SO, obj, *objs])
    print("built", SO)
    sys.exit(0)

import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.zeros(16 * 2 ** 20, dtype=torch.float32, device=dev)
cold = torch.empty(160 * 2 ** 20, dtype=torch.float32, device=dev)
TILE = {13: (128, 128), 14: (128, 64), 15: (64, 64)}


def report(name, launch, nblk, prep):
    for _ in range(2):
        launch()
    rows = []
    for rep in range(5):
        prep()
        ws.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        launch()
        e.record()
        torch.cuda.synchronize()
        st = ws.view(torch.int64)[: nblk * 8].view(nblk, 8).cpu().double()
        t0 = st[:, 0].min()
        d = lambda a, b: float((st[:, b] - st[:, a]).median())
        rows.append([float(st[:, 0].max() - t0), d(0, 1), d(1, 2), d(2, 3), d(3, 4), d(4, 5), d(5, 6), float(st[:, 6].max() - t0),
                     s.elapsed_time(e) * 1e3])
    r = torch.tensor(rows).median(0).values.tolist()
    tick = 0.01  # s_memtime: 100 MHz
    print(f"{name}: {nblk} blocks | entry skew {r[0] * tick:5.2f}  prologue {r[1] * tick:5.2f}  first stage lands {r[2] * tick:5.2f}  "
          f"K loop {r[3] * tick:5.2f}  acc->LDS {r[4] * tick:5.2f}  epilogue stores {r[5] * tick:5.2f}  drain {r[6] * tick:5.2f}  "
          f"| first entry -> last exit {r[7] * tick:5.2f} us, events {r[8]:5.1f} us", flush=True)


for (M, N, K, tile, resid) in [(4096, 640, 640, 13, False), (4096, 640, 640, 13, True), (16384, 320, 320, 13, True), (1024, 1280, 1280, 15, True),
                               (4096, 640, 2560, 13, True), (4928, 768, 768, 13, False)]:
    A = torch.randn(M, K, device=dev).half()
    B = (torch.randn(N, K, device=dev) * 0.03).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    R = torch.randn(M, N, device=dev).half() if resid else None
    bias = torch.randn(N, device=dev)
    bm, bn = TILE[tile]
    nblk = -(-M // bm) * -(-N // bn)

    def prep():
        cold.fill_(0)
        A.add_(0)
        if R is not None:
            R.add_(0)

    report(f"cold  {M}x{N}x{K} tile {tile} resid={resid}", lambda: ops.gemm(A, B, C, bias=bias, resid=R, tile_hint=tile, split_k=1, workspace=ws), nblk, prep)
    report(f"hot   {M}x{N}x{K} tile {tile} resid={resid}", lambda: ops.gemm(A, B, C, bias=bias, resid=R, tile_hint=tile, split_k=1, workspace=ws), nblk, lambda: None)
