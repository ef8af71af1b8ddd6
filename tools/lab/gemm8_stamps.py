"""[needs the lab switches: git apply tools/lab/attic/lab_switches.patch first — tools/lab/README.md]
Lab: where a 256x256 8-phase tile spends its life.  Builds csrc/gemm8.hip with -DVN_GEMM8_STAMP into
tools/lab/libvneti_stamp.so (thread 0 of every block records s_memtime at: entry, first data landed, loop end, C tile
in LDS, stores issued, stores acknowledged) and prints the per-section medians over the blocks of one launch.
    python tools/lab/gemm8_stamps.py build          (in the container)
    VNETI_LIB_PATH=tools/lab/libvneti_stamp.so python tools/lab/gemm8_stamps.py   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CS = os.path.join(ROOT, "view_neti_amd", "csrc")
SO = os.path.join(ROOT, "tools", "lab", "libvneti_stamp.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    obj = "/tmp/gemm8_stamp.o"
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-fno-fast-math",
                           "-DVN_GEMM8_STAMP", "-c", os.path.join(CS, "gemm8.hip"), "-o", obj])
    objs = [os.path.join(CS, "build", f) for f in os.listdir(os.path.join(CS, "build")) if f.endswith(".o") and f != "gemm8.o"]
    subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", SO, obj, *objs])
    print("built", SO)
    sys.exit(0)

import torch
from view_neti_amd import ops
dev = "cuda"
ws = torch.zeros(64 * 2 ** 20 // 4, dtype=torch.float32, device=dev)


def report(name, launch, nblk):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    ws.zero_()
    launch()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[: nblk * 8].view(nblk, 8).cpu().double()
    t0 = st[:, 0].min()
    d = lambda a, b: (st[:, b] - st[:, a])
    med = lambda x: float(x.median())
    clk = 100.0  # s_memtime ticks at a constant 100 MHz on this part?  calibrated below against the launch duration
    span = float(st[:, 5].max() - t0)
    print(f"{name}: {nblk} blocks; ticks: launch span {span:.0f}; per block median: prologue {med(d(0, 1)):.0f}  loop {med(d(1, 2)):.0f}  "
          f"acc->LDS {med(d(2, 3)):.0f}  stores issued {med(d(3, 4)):.0f}  store drain {med(d(4, 5)):.0f}  total {med(d(0, 5)):.0f}; "
          f"start skew (max entry - min entry) {float(st[:, 0].max() - t0):.0f}")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        launch()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 5 * 1e3
    print(f"    launch {us:.1f} us  => {span / us:.1f} ticks per us")


for (M, N, K) in [(4096, 4096, 64), (4096, 4096, 4096), (65536, 512, 4608)]:
    A = torch.randn(M, K, device=dev).half()
    B = torch.randn(N, K, device=dev).half()
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    report(f"gemm {M}x{N}x{K}", lambda: ops.gemm(A, B, C, tile_hint=16, split_k=1, workspace=ws), (M // 256) * (N // 256))
for (Bn, H, W, Ci, Co) in [(4, 256, 256, 256, 256), (4, 128, 128, 512, 512), (4, 64, 64, 640, 640)]:
    x = torch.randn(Bn * H * W, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
    y = torch.empty(Bn * H * W, Co, device=dev, dtype=torch.float16)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci)
    report(f"conv {H}x{W} {Ci}->{Co}", lambda: ops.gemm(x, w, y, M=Bn * H * W, conv=conv, tile_hint=16, workspace=ws, split_k=1),
           (Bn * H * W // 256) * ((Co + 255) // 256))
# the N = 128 convolution of the VAE: row-major 256x128 tile (17) against its halo-patch form (18, chunk-major K)
for hint in (17, 18):
    Bn, H, W, Ci, Co = 4, 512, 512, 128, 128
    x = torch.randn(Bn * H * W, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
    y = torch.empty(Bn * H * W, Co, device=dev, dtype=torch.float16)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1 if hint == 18 else 0)
    conv["korder"] = 0 if hint == 17 else 1
    report(f"conv {H}x{W} {Ci}->{Co} tile {hint}", lambda: ops.gemm(x, w, y, M=Bn * H * W, conv=conv, tile_hint=hint, workspace=ws, split_k=1),
           Bn * H * W // (512 if hint == 19 else 256))
