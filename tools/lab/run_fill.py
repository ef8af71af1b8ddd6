import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
MODES = {0: "LDS-DMA b128", 1: "load->VGPR->ds_write_b128", 2: "load->VGPR only", 3: "LDS-DMA b32"}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for m in MODES:
        r = subprocess.run(f"hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DMODE={m} {HERE}/fill_lab.hip -o {HERE}/fill_{m}.so", shell=True, capture_output=True, text=True)
        print(m, "ok" if r.returncode == 0 else r.stderr[-1500:])
    sys.exit(0)
import torch
out = torch.zeros(4096, device="cuda", dtype=torch.int32)
src = torch.randint(0, 2**31 - 1, (64 * 1024 * 1024,), device="cuda", dtype=torch.int32)  # 256 MB
for m, name in MODES.items():
    lib = ctypes.CDLL(f"{HERE}/fill_{m}.so")
    lib.fill_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
    for label, span, stride in (("shared 256KB (L2-hot, all CUs same)", 256 * 1024, 0), ("private 64KB per CU (L1/L2-hot)", 64 * 1024, 64 * 1024),
                                ("private 1MB per CU (HBM/MALL stream)", 1024 * 1024, 1024 * 1024)):
        for waves in (4, 8, 16):
            iters = 2000
            st = torch.cuda.current_stream().cuda_stream
            lib.fill_launch(src.data_ptr(), out.data_ptr(), 256, waves * 64, 20, span, stride, st)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            lib.fill_launch(src.data_ptr(), out.data_ptr(), 256, waves * 64, iters, span, stride, st)
            e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) * 1e-3
            nbytes = 256.0 * waves * 8 * 1024 * iters
            print(f"{name:28s} {label:38s} waves {waves:2d}: {nbytes/t/1e12:6.2f} TB/s = {nbytes/t/256/2.4e9:5.1f} B/clk/CU")
