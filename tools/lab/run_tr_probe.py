import ctypes, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(f"{HERE}/tr_probe.so")
lib.tr_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
for mode in range(4):
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    lib.tr_launch(out.data_ptr(), mode, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().view(64, 4).tolist()
    print("mode", mode)
    for l in (0, 1, 2, 3, 4, 5, 15, 16, 17, 31, 32, 48, 63):
        print(f"  lane {l:2d}: {o[l]}")
