"""Build + check + time the halo-patch conv lab kernel against the production implicit-GEMM conv.
usage: python tools/lab/run_halo.py build   (here, cross-compile)   |   python tools/lab/run_halo.py   (GPU box)"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
VARIANTS = {
    "16x16_n128": "-DTH=16 -DTW=16 -DBN=128",
    "16x16_n128_p20": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20",
    "8x16_n256": "-DTH=8 -DTW=16 -DBN=256",
    "8x16_n128": "-DTH=8 -DTW=16 -DBN=128",
    "16x16_n64": "-DTH=16 -DTW=16 -DBN=64 -DWN=32",
    "full_16x16_n128_p20": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DFULLSET",
    "full_abl_nodma_nobar": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DFULLSET -DABL_NODMA -DABL_NOBARRIER",
    "full_8x16_n256": "-DTH=8 -DTW=16 -DBN=256 -DFULLSET",
    "abl_nodma": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DABL_NODMA",
    "abl_nomfma": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DABL_NOMFMA",
    "abl_nolds": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DABL_NOLDS",
    "abl_nobar": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DABL_NOBARRIER",
    "abl_nodma_nobar": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DABL_NODMA -DABL_NOBARRIER",
    "abl_nodma_nolds": "-DTH=16 -DTW=16 -DBN=128 -DPITCH=20 -DABL_NODMA -DABL_NOLDS",
}
ONLY = os.environ.get("ONLY")
if ONLY:
    VARIANTS = {k: v for k, v in VARIANTS.items() if any(o in k for o in ONLY.split(","))}
def build():
    for name, flags in VARIANTS.items():
        so = os.path.join(HERE, f"halo_{name}.so")
        cmd = f"hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC {flags} {HERE}/conv_halo_lab.hip -o {so} -Rpass-analysis=kernel-resource-usage"
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True)
        info = [l.split("remark:")[1].strip() for l in r.stderr.splitlines() if any(k in l for k in ("VGPRs:", "ScratchSize", "Occupancy", "LDS Size"))]
        print(name, "ok" if r.returncode == 0 else r.stderr[-800:], " | ".join(i.split("[-R")[0] for i in info))
def run():
    import torch
    from view_neti_amd import ops
    ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device="cuda")
    shapes = [(4, 512, 512, 128, 128)] if os.environ.get("ONLY") else [(4, 512, 512, 128, 128), (4, 256, 256, 256, 256), (4, 128, 128, 512, 512), (4, 64, 64, 320, 320), (4, 64, 64, 640, 320),
              (4, 32, 32, 640, 640), (4, 32, 32, 1280, 640), (4, 16, 16, 1280, 1280)]
    for (B, H, W, Ci, N) in shapes:
        x = torch.randn(B * H * W, Ci, device="cuda").half()
        w = (torch.randn(N, 9 * Ci, device="cuda") * (1.0 / (9 * Ci) ** 0.5)).half()
        y0 = torch.empty(B * H * W, N, device="cuda", dtype=torch.float16)
        conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci)
        best = (None, 1e9)
        for h in (1, 2, 5, 6):
            for sk in (0, 1):
                f = lambda: ops.gemm(x, w, y0, M=B * H * W, conv=conv, tile_hint=h, workspace=ws, split_k=sk)
                f(); f()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5): f()
                e.record(); torch.cuda.synchronize()
                t = s.elapsed_time(e) / 5 * 1e3
                if t < best[1]: best = ((h, sk), t)
        ops.gemm(x, w, y0, M=B * H * W, conv=conv, tile_hint=best[0][0], workspace=ws, split_k=best[0][1])
        gf = 2.0 * B * H * W * N * 9 * Ci / 1e9
        line = f"B{B} {H}x{W} {Ci}->{N} {gf:7.1f}GF | prod best h{best[0]} {best[1]:8.1f}us {gf/best[1]*1e3:5.0f}TF"
        for name in VARIANTS:
            so = os.path.join(HERE, f"halo_{name}.so")
            if not os.path.exists(so): continue
            lib = ctypes.CDLL(so)
            lib.halo_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
            y = torch.zeros_like(y0)
            st = torch.cuda.current_stream().cuda_stream
            rc = lib.halo_launch(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, Ci, N, st)
            if rc != 0:
                line += f" | {name}: n/a"; continue
            torch.cuda.synchronize()
            err = ((y.float() - y0.float()).norm() / y0.float().norm()).item()
            for _ in range(2): lib.halo_launch(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, Ci, N, st)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): lib.halo_launch(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, Ci, N, st)
            e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) / 5 * 1e3
            line += f" | {name}: {t:7.1f}us {gf/t*1e3:5.0f}TF" + ("" if err < 2e-3 else f" !ERR{err:.1e}")
        print(line, flush=True)
if __name__ == "__main__":
    (build if len(sys.argv) > 1 and sys.argv[1] == "build" else run)()
