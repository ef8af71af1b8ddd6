"""Lab: the direct conv_in kernel alone (4 x 3 x 512 x 512 f32 -> [1M][128] f16, GroupNorm sums on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops, packing
dev = "cuda"
B, H, W, Co = 4, 512, 512, 128
x = torch.randn(B, 3, H, W, device=dev)
w = packing.conv_in_direct(torch.randn(Co, 3, 3, 3) * 0.2).half().to(dev)
bias = torch.randn(Co, device=dev)
out = torch.empty(B * H * W, Co, dtype=torch.float16, device=dev)
sums = torch.zeros(B, 8, 32, 2, device=dev)
f = lambda: ops.conv3x3_in(x, w, bias, out, B, 3, H, W, x.stride(), gn_sums=sums, gn_hw=H * W, gn_groups=32, gn_slots=8)
for _ in range(3): f()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10
print(f"conv_in direct: {t*1e3:.1f} us, {out.numel()*2/t/1e9:.2f} TB/s of output")
