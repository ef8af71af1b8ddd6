import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "build":
    for nacc in (2, 4, 8):
        r = subprocess.run(f"hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DNACC={nacc} {HERE}/mfma_peak.hip -o {HERE}/mfma_peak_{nacc}.so", shell=True, capture_output=True, text=True)
        print(nacc, "ok" if r.returncode == 0 else r.stderr[-400:])
    sys.exit(0)
import torch
out = torch.zeros(4096, device="cuda")
for nacc in (2, 4, 8):
    lib = ctypes.CDLL(f"{HERE}/mfma_peak_{nacc}.so")
    lib.mfma_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for waves in (4, 8, 16):
        for blocks_per_cu in (1, 2):
            iters = 20000
            blocks = 256 * blocks_per_cu
            st = torch.cuda.current_stream().cuda_stream
            lib.mfma_launch(out.data_ptr(), blocks, waves * 64, 100, st)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            lib.mfma_launch(out.data_ptr(), blocks, waves * 64, iters, st)
            e.record(); torch.cuda.synchronize()
            t = s.elapsed_time(e) * 1e-3
            flops = blocks * waves * iters * nacc * 32768.0
            print(f"nacc {nacc} waves/block {waves:2d} blocks/CU {blocks_per_cu}: {t*1e3:7.2f} ms  {flops/t/1e12:7.0f} TF/s")
