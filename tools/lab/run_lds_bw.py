import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for m in (0, 1, 2):
        r = subprocess.run(f"hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DMODE={m} {HERE}/lds_bw.hip -o {HERE}/lds_bw_{m}.so", shell=True, capture_output=True, text=True)
        print(m, "ok" if r.returncode == 0 else r.stderr[-400:])
    sys.exit(0)
import torch
out = torch.zeros(4096, device="cuda", dtype=torch.int32)
for m, name in ((0, "swizzled frag pattern"), (1, "lane-linear"), (2, "broadcast")):
    lib = ctypes.CDLL(f"{HERE}/lds_bw_{m}.so")
    lib.lds_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for waves in (4, 8, 16):
        iters = 4000
        st = torch.cuda.current_stream().cuda_stream
        lib.lds_launch(out.data_ptr(), 256, waves * 64, 50, st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        lib.lds_launch(out.data_ptr(), 256, waves * 64, iters, st)
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) * 1e-3
        nbytes = 256.0 * waves * iters * 16 * 1024
        print(f"{name:24s} waves/CU {waves:2d}: {nbytes/t/1e12:6.1f} TB/s aggregate = {nbytes/t/256/1e9:6.0f} GB/s/CU = {nbytes/t/256/2.4e9:5.1f} B/clk/CU @2.4GHz")
