"""Build + time the GEMM lab variants.  usage: python tools/lab/run_lab.py (builds here, run on GPU box)"""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {
    "128x128": "",
    "128x128_fragdb": "-DFRAGDB",
    "128x128_fragdb_prio": "-DFRAGDB -DSETPRIO",
    "128x64_fragdb": "-DBM=128 -DBN=64 -DWM=64 -DWN=32 -DFRAGDB",
    "256x128_8w": "-DBM=256 -DBN=128",
    "256x128_8w_fragdb": "-DBM=256 -DBN=128 -DFRAGDB",
    "256x256_16w": "-DBM=256 -DBN=256",
    "256x256_16w_fragdb": "-DBM=256 -DBN=256 -DFRAGDB",
    "256x256_8w_128x64": "-DBM=256 -DBN=256 -DWM=128 -DWN=64",
    "256x256_8w_128x64_fragdb": "-DBM=256 -DBN=256 -DWM=128 -DWN=64 -DFRAGDB",
    "256x256_8w_128x64_fragdb_prio": "-DBM=256 -DBN=256 -DWM=128 -DWN=64 -DFRAGDB -DSETPRIO",
    "256x128_4w_128x64_fragdb": "-DBM=256 -DBN=128 -DWM=128 -DWN=64 -DFRAGDB",
    "256x256_16w_xpf32x3": "-DBM=256 -DBN=256 -DXPF -DXBK=32 -DXST=3",
    "256x256_16w_xpf32x4": "-DBM=256 -DBN=256 -DXPF -DXBK=32 -DXST=4",
    "256x128_8w_xpf64x3": "-DBM=256 -DBN=128 -DXPF -DXBK=64 -DXST=3",
    "256x128_8w_xpf32x4": "-DBM=256 -DBN=128 -DXPF -DXBK=32 -DXST=4",
    "128x128_xpf32x4": "-DXPF -DXBK=32 -DXST=4",
    "128x128_xpf64x3": "-DXPF -DXBK=64 -DXST=3",
}
ONLY = os.environ.get("ONLY")
if ONLY:
    VARIANTS = {k: v for k, v in VARIANTS.items() if any(o in k for o in ONLY.split(","))}
def build():
    for name, flags in VARIANTS.items():
        so = os.path.join(HERE, f"lab_{name}.so")
        cmd = f"hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC {flags} {HERE}/gemm_lab.hip -o {so}"
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True)
        print(name, "ok" if r.returncode == 0 else r.stderr[-600:])
def run():
    import torch
    shapes = [(4096, 4096, 4096), (16384, 640, 5760), (16384, 320, 2880), (4096, 1280, 11520), (65536, 512, 4608), (4928, 3072, 768)]
    for name in VARIANTS:
        so = os.path.join(HERE, f"lab_{name}.so")
        if not os.path.exists(so):
            continue
        lib = ctypes.CDLL(so)
        lib.lab_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
        line = f"{name:24s}"
        for (M, N, K) in shapes:
            A = (torch.rand(M, K, device="cuda") * 2 - 1).half(); B = (torch.rand(N, K, device="cuda") * 2 - 1).half()
            C = torch.empty(M, N, dtype=torch.float16, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3): lib.lab_launch(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): lib.lab_launch(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)
            e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) / 10
            if True:
                ref = (A[:64].float() @ B[:64].float().t()); err = (C[:64, :64].float() - ref).abs().max().item()
                ok = "" if err < 0.5 else f"!ERR{err:.1f}"
            else:
                ok = ""
            line += f" | {M}x{N}x{K}: {2.0*M*N*K/t/1e9:6.0f}TF{ok}"
        print(line)
if __name__ == "__main__":
    (build if len(sys.argv) > 1 and sys.argv[1] == "build" else run)()
