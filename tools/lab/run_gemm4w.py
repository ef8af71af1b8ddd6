"""Lab: the 4-wave / 128x128-wave-tile GEMM (tools/lab/gemm4w_lab.hip; build: hipcc -shared, see below) against the 8-phase
256x256 tile (tile_hint 16), the generic 256x256 tile (5) and hipBLASLt (torch.matmul), random f16 operands, interleaved.
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -shared -o tools/lab/libgemm4w.so tools/lab/gemm4w_lab.hip
    python tools/lab/run_gemm4w.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops

lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm4w.so"))
lab.lab_gemm4w.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
ops.set_default_gemm_workspace(torch.empty(64 * 2 ** 20, dtype=torch.float32, device="cuda"))


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (M, N, K) in [(512, 512, 256), (4096, 4096, 4096), (8192, 8192, 4096), (65536, 512, 4608), (16384, 768, 5760), (4864, 3072, 768)]:
    g = torch.Generator().manual_seed(M + K)
    A = (torch.rand(M, K, generator=g) * 2 - 1).half().cuda()
    B = (torch.rand(N, K, generator=g) * 2 - 1).half().cuda()
    C = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lab.lab_gemm4w(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)
    torch.cuda.synchronize()
    C16 = torch.empty_like(C)
    ops.gemm(A, B, C16, tile_hint=16, split_k=1)
    torch.cuda.synchronize()
    same = torch.equal(C, C16)
    ref = A[:512].float() @ B.float().t()
    err = ((C[:512].float() - ref).norm() / ref.norm()).item()
    res = {}
    for rnd in range(3):  # interleaved
        res.setdefault("4w", []).append(timeit(lambda: lab.lab_gemm4w(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, st)))
        res.setdefault("t16", []).append(timeit(lambda: ops.gemm(A, B, C16, tile_hint=16, split_k=1)))
        res.setdefault("t5", []).append(timeit(lambda: ops.gemm(A, B, C16, tile_hint=5, split_k=1)))
        res.setdefault("blas", []).append(timeit(lambda: torch.matmul(A, B.t())))
    tf = lambda us: 2.0 * M * N * K / us / 1e6
    print(f"{M}x{N}x{K}: rc {rc} rel err {err:.2e} bit-equal to tile 16: {same} | TF/s  " +
          "  ".join(f"{k} {min(tf(x) for x in v):.0f}-{max(tf(x) for x in v):.0f}" for k, v in res.items()), flush=True)
