"""Lab: s_memtime stamps inside the halo loop (tools/lab/libvneti_hstamp.so, a build with HST() stamps) for blocks 1000 /
1001, K-tiles 6..9 of the 512^2 128->128 conv: per wave the eight section boundaries of each K-tile
  0 loop top | 1 Q0 loads issued | 2 past Q0 barrier | 3 Q0 MFMAs issued | 4 past phase-end barrier | 5 Q1 reads issued
  (+ vmcnt(3)) | 6 past Q1 sync | 7 Q1 MFMAs issued
    VNETI_LIB_PATH=tools/lab/libvneti_hstamp.so python tools/lab/halo_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
dev = "cuda"
B, H, W, Ci, Co = 4, 512, 512, 128, 128
x = torch.randn(B * H * W, Ci, device=dev).half()
w = (torch.randn(Co, 9 * Ci, device=dev) * 0.03).half()
y = torch.empty(B * H * W, Co, device=dev, dtype=torch.float16)
ws = torch.zeros(16 * 2 ** 20, dtype=torch.float32, device=dev)
conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
for _ in range(3):
    ops.gemm(x, w, y, M=B * H * W, conv=conv, tile_hint=18, workspace=ws, split_k=1)
torch.cuda.synchronize()
st = ws.view(torch.int64)[: 2 * 8 * 4 * 8].cpu().view(2, 8, 4, 8)
for blk in range(2):
    t0 = int(st[blk, :, 0, 0].min())
    print(f"block {1000 + blk}: cycles relative to the earliest wave's K-tile-6 top")
    for wave in range(8):
        row = []
        for t in range(4):
            row.append(" ".join(f"{int(st[blk, wave, t, k]) - t0:6d}" for k in range(8)))
        print(f" wave {wave} (row {wave >> 2}): " + " | ".join(row))
    # interval lengths between consecutive barriers as seen by wave 0 and wave 4
    for wave in (0, 4):
        d = []
        for t in range(4):
            s = [int(v) for v in st[blk, wave, t]]
            d.append(f"loadsQ0 {s[1]-s[0]:4d} bar {s[2]-s[1]:4d} mfma {s[3]-s[2]:4d} bar {s[4]-s[3]:4d} readsQ1 {s[5]-s[4]:4d} sync {s[6]-s[5]:4d} mfma {s[7]-s[6]:4d}")
        print(f"  wave {wave}: " + " || ".join(d))
    print(f"  K-tile period (wave 0): {[int(st[blk,0,t+1,0]-st[blk,0,t,0]) for t in range(3)]}")
