import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {0: "reads only", 1: "mfma only", 2: "read-all/wait/mfma-all", 3: "software pipeline", 4: "independent interleave", 5: "mfma 16x16x32 only (same flops)"}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for m in MODES:
        for tag, fl in (("", ""), ("r", "-DRANDOM_DATA")):
            r = subprocess.run(f"hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DMODE={m} {fl} {HERE}/overlap_lab.hip -o {HERE}/overlap_{m}{tag}.so", shell=True, capture_output=True, text=True)
            print(m, tag, "ok" if r.returncode == 0 else r.stderr[-1500:])
    sys.exit(0)
import torch
out = torch.zeros(4096, device="cuda", dtype=torch.float32)
TAG = "r" if len(sys.argv) > 1 and sys.argv[1] == "random" else ""
ONLY = [int(x) for x in os.environ.get("MODES", "0,1,2,3,4,5").split(",")]
for m, name in MODES.items():
    if m not in ONLY: continue
    lib = ctypes.CDLL(f"{HERE}/overlap_{m}{TAG}.so")
    lib.ovl_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for waves in (4, 8, 16):
        iters = 200000 if TAG else 4000
        st = torch.cuda.current_stream().cuda_stream
        lib.ovl_launch(out.data_ptr(), 256, waves * 64, 50, st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        lib.ovl_launch(out.data_ptr(), 256, waves * 64, iters, st)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e-3
        clk = t / iters * 2.4e9
        print(f"{name:26s} waves/CU {waves:2d}: {clk:7.0f} clk/iter @2.4GHz  (mfma ideal {16*32*waves//4}, lds ideal {16*1024*waves//256})")
