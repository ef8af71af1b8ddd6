"""Dev tool: one implicit-GEMM conv problem (default: the VAE's 512x512x128 3x3 conv) launched a few times, to be
wrapped in rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import ops
B, H, W = 4, int(os.environ.get("HW", 512)), int(os.environ.get("HW", 512))
Ci, Co = int(os.environ.get("CI", 128)), int(os.environ.get("CO", 128))
hint = int(os.environ.get("HINT", 6))
dev = "cuda"
x = torch.randn(B * H * W, Ci, device=dev).half()
w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
ws = torch.empty(16 * 2 ** 20, dtype=torch.float32, device=dev)
conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci,
            korder=int(os.environ.get("KORDER", 0)))  # (timing / counters only: the random weight has no K order)
for _ in range(int(os.environ.get("REPS", 3))):
    ops.gemm(x, w, y, M=B * H * W, conv=conv, tile_hint=hint, workspace=ws, split_k=1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3):
    ops.gemm(x, w, y, M=B * H * W, conv=conv, tile_hint=hint, workspace=ws, split_k=1)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 3
print(f"conv {H}x{W} Ci={Ci} Co={Co} hint {hint}: {t*1e3:.1f} us  {2.0*B*H*W*Co*9*Ci/t/1e9:.0f} TF/s")
