"""Lab: tile 13 (persistent form, lab library via VNETI_LIB_PATH) must be bit-identical to tile 8 (same block shape, 2-stage
ring, one tile per block); GroupNorm sums equal to f32 summation order."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from view_neti_amd import ops
DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


for variant in ("plain", "conv", "rowadd_resid_act", "gn4", "gn8_ragged"):
    G, S = 32, 8
    N = 256 if variant == "gn8_ragged" else 128
    Bn, HW = (3, 65536) if variant != "gn8_ragged" else (2, 256 * 350)
    M = Bn * HW if variant != "plain" else Bn * HW - 77
    K = 128
    kw = {}
    B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=72)
    if variant == "conv":
        Bn, H, W, Ci = 2, 256, 260, 64
        M, K = Bn * H * W, 9 * 64
        A = rnd(Bn, H, W, Ci, seed=71).to(DEV).view(-1, Ci)
        B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=72)
        kw = dict(conv=dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci), M=M,
                  bias=(rnd(N, seed=73, dtype=torch.float32) * 2).to(DEV))
    else:
        A = rnd(M, K, seed=71).to(DEV)
    if variant == "rowadd_resid_act":
        kw = dict(bias=(rnd(N, seed=73, dtype=torch.float32) * 2).to(DEV), act=1, rowadd=rnd(Bn, N, seed=74).to(DEV),
                  rows_per_group=HW, resid=rnd(M, N + 8, seed=75).to(DEV)[:, :N], alpha=0.7)
    outs, sums = [], []
    for hint in (13, 8):
        out = torch.zeros(M, N + 8, dtype=torch.float16, device=DEV)[:, :N]
        k2 = dict(kw)
        if variant.startswith("gn"):
            sm = torch.zeros(Bn, S, G, 2, dtype=torch.float32, device=DEV)
            k2.update(bias=(rnd(N, seed=73, dtype=torch.float32) * 2).to(DEV), gn_sums=sm, gn_hw=HW, gn_groups=G, gn_slots=S)
            sums.append(sm)
        ops.gemm(A, B.to(DEV), out, tile_hint=hint, split_k=1, **k2)
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]), variant
    if sums:
        a, b = sums[0].sum(1), sums[1].sum(1)
        assert float((a - b).abs().max() / b.abs().max()) < 1e-5, variant
    print(f"tile 13 == tile 8: {variant}")
