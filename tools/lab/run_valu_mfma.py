"""Lab driver: VALU / MFMA overlap on one SIMD (valu_mfma_lab.hip).  `build` compiles (no GPU needed)."""
import ctypes, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = f"{HERE}/valu_mfma_lab.so"
if len(sys.argv) > 1 and sys.argv[1] == "build":
    r = subprocess.run(f"hipcc --offload-arch=gfx950 -O3 -shared -fPIC {HERE}/valu_mfma_lab.hip -o {SO}", shell=True, capture_output=True, text=True)
    print("ok" if r.returncode == 0 else r.stderr[-3000:]); sys.exit(r.returncode)
import torch
MODES = {0: "mfma only", 1: "valu only", 2: "mfma block; valu block", 3: "fine interleave", 4: "wave-specialised",
         5: "phase-shifted wave groups", 6: "as 2 + setprio around mfma"}
lib = ctypes.CDLL(SO)
lib.vm_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(4096, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for waves in (8, 16):
    for m, name in MODES.items():
        iters = 20000
        lib.vm_launch(out.data_ptr(), 256, waves * 64, 200, m, st); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); lib.vm_launch(out.data_ptr(), 256, waves * 64, iters, m, st); e.record(); torch.cuda.synchronize()
        ns = s.elapsed_time(e) * 1e6 / iters
        print(f"waves/SIMD {waves//4}  mode {m} {name:32s} {ns:8.1f} ns/iter  = {ns/ (waves//4):7.1f} ns per wave-iteration per SIMD")
