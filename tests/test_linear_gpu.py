"""The row-stationary persistent linear kernel (csrc/linear.hip, tile_hint 19 of vneti_gemm_f16) at the shapes the train
step gives it: the transformer blocks' projections (models/xti_attention_processor.py:30-55, diffusers
BasicTransformerBlock) and the CLIP MLP.  Three things are pinned:
  * it is BIT-IDENTICAL to the tiled kernels on every epilogue it offers (same MFMA shape, same k order inside a row, same
    rounding points) — so switching a launch to it changes nothing downstream;
  * the fused LayerNorm prologue equals vneti_layernorm_fwd followed by the GEMM up to the last bit of the row statistics
    (eight lanes sum a row instead of sixty-four), and publishes the same mean / rstd;
  * ineligible problems fall back to the heuristic tile and a fused LayerNorm on one is an error, not a silent change."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


SHAPES = [(16384, 320, 320), (4096, 640, 640), (4928, 768, 768), (16384, 960, 320), (4096, 1920, 640), (4928, 2304, 768),
          (1000, 328, 192), (77, 3072, 768), (64, 16, 64), (16384, 2560, 320)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", ["plain", "bias+resid"])
def test_linear_equals_tiled_kernel_bit_for_bit(M, N, K, epi):
    from view_neti_amd import ops
    A = rnd(M, K, seed=1).to(DEV)
    B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=2).to(DEV)
    kw = {}
    if epi != "plain":
        kw = dict(bias=rnd(N, seed=3, dtype=torch.float32).to(DEV), resid=rnd(M, N, seed=4).to(DEV))
    want = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    got = torch.full((M, N), 7.0, dtype=torch.float16, device=DEV)
    ops.gemm(A, B, want, tile_hint=13, split_k=1, **kw)
    ops.gemm(A, B, got, tile_hint=19, **kw)
    torch.cuda.synchronize()
    ref = A.float().cpu() @ B.float().cpu().t()
    if kw:
        ref = (ref + kw["bias"].cpu()).half().float() + kw["resid"].float().cpu()
    rel = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    assert rel < 2e-3, rel
    assert torch.equal(got, want), f"tile 19 differs from the tiled kernel in {(got != want).sum().item()} elements"


@pytest.mark.parametrize("M,C", [(16384, 320), (4096, 640), (300, 64)])
def test_linear_geglu_and_gate_epilogues_equal_tiled(M, C):
    from view_neti_amd import ops, packing
    F4 = 4 * C
    x = rnd(M, C, seed=61).to(DEV)
    W = packing.geglu_interleave(rnd(2 * F4, C, scale=1.0 / math.sqrt(C), seed=62)).to(DEV)
    b = packing.geglu_interleave(rnd(2 * F4, seed=63, dtype=torch.float32)).to(DEV)
    outs = {}
    for hint in (9, 19):
        p = torch.zeros(M, 2 * F4, dtype=torch.float16, device=DEV)
        gg = torch.zeros(M, F4, dtype=torch.float16, device=DEV)
        ops.gemm(x, W, p, bias=b, out2=gg, geglu=1, split_k=1, tile_hint=hint)
        dy = rnd(M, C, seed=64).to(DEV)
        W2t = rnd(F4, C, scale=1.0 / math.sqrt(F4), seed=65).to(DEV)  # ff.net.2.weight^T: [4C][C]
        dp = torch.zeros(M, 2 * F4, dtype=torch.float16, device=DEV)
        ops.gemm(dy, W2t, dp, gate=p, gate_act=ops.ACT_GELU, geglu=2, split_k=1, tile_hint=hint)
        # the CLIP MLP pair: fc1 with a second activated output, and the activation-gradient gate of its backward
        h = torch.zeros(M, F4, dtype=torch.float16, device=DEV)
        a = torch.zeros(M, F4, dtype=torch.float16, device=DEV)
        ops.gemm(x, W[:F4], h, bias=b[:F4], out2=a, act2=ops.ACT_QUICK_GELU, tile_hint=hint, split_k=1)
        dh = torch.zeros(M, F4, dtype=torch.float16, device=DEV)
        ops.gemm(dy, W2t, dh, gate=h, gate_act=ops.ACT_QUICK_GELU, tile_hint=hint, split_k=1)
        torch.cuda.synchronize()
        outs[hint] = (p, gg, dp, h, a, dh)
    for name, u, v in zip(("pre-activation", "h*gelu(g)", "geglu backward", "fc1", "act2(fc1)", "gated dgrad"), outs[9], outs[19]):
        assert torch.isfinite(v.float()).all() and v.float().abs().max() > 0, name
        assert torch.equal(u, v), f"{name}: tile 19 differs from the tiled kernel in {(u != v).sum().item()} elements"


@pytest.mark.parametrize("M,N,K", [(16384, 960, 320), (4096, 640, 640), (16384, 2560, 320), (1000, 328, 192), (70, 64, 768)])
def test_linear_fused_layernorm(M, N, K):
    from view_neti_amd import ops, packing
    x = (rnd(M, K, seed=24, dtype=torch.float32) * 2 + 0.5).half().to(DEV)
    gamma = (1 + 0.1 * rnd(K, seed=25, dtype=torch.float32)).to(DEV)
    beta = (0.1 * rnd(K, seed=26, dtype=torch.float32)).to(DEV)
    W = rnd(N, K, scale=1.0 / math.sqrt(K), seed=27).to(DEV)
    bias = rnd(N, seed=28, dtype=torch.float32).to(DEV)
    geglu = N == 2560
    y = torch.zeros(M, K, dtype=torch.float16, device=DEV)
    mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
    ops.layernorm_fwd(x, y, gamma, beta, mean, rstd, 1e-5)
    want = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    got = torch.zeros_like(want)
    mean2, rstd2 = torch.full((M,), 9.0, device=DEV), torch.full((M,), 9.0, device=DEV)
    kw, kw2 = {}, {}
    if geglu:
        W, bias = packing.geglu_interleave(W.cpu()).to(DEV), packing.geglu_interleave(bias.cpu()).to(DEV)
        kw = dict(out2=torch.zeros(M, N // 2, dtype=torch.float16, device=DEV), geglu=1)
        kw2 = dict(out2=torch.zeros(M, N // 2, dtype=torch.float16, device=DEV), geglu=1)
    ops.gemm(y, W, want, bias=bias, tile_hint=19, **kw)
    ops.gemm(x, W, got, bias=bias, tile_hint=19, ln=(gamma, beta, mean2, rstd2, 1e-5), **kw2)
    torch.cuda.synchronize()
    # the statistics: same two-pass arithmetic, a different summation tree
    assert torch.allclose(mean2, mean, rtol=0, atol=2e-6 * float(x.float().abs().max()))
    assert torch.allclose(rstd2, rstd, rtol=2e-6, atol=0)
    ref = F.layer_norm(x.float().cpu(), (K,), gamma.cpu(), beta.cpu(), 1e-5).half().float() @ W.float().cpu().t() + bias.cpu()
    rel = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    # a last-bit difference in mean / rstd flips the f16 rounding of a few normalised values: ~1e-4 of the elements move by
    # one f16 ulp of the product — far below the tolerance against the fp32 reference, and never more than that
    diff = (got.float() - want.float()).abs()
    frac = (diff > 0).float().mean().item()
    print(f"[fused LN {M}x{N}x{K}] rel vs fp32 reference {rel:.2e}; differs from LN-kernel + GEMM in {100 * frac:.3f}% of the "
          f"outputs, max |diff| {diff.max().item():.3e}")
    assert rel < 2e-3 and frac < 0.02 and diff.max().item() <= 4e-3 * float(want.float().abs().max())
    if geglu:
        d2 = (kw["out2"].float() - kw2["out2"].float()).abs()
        assert (d2 > 0).float().mean().item() < 0.02


def test_linear_fallback_and_errors():
    from view_neti_amd import ops
    A = rnd(256, 1280, seed=1).to(DEV)
    B = rnd(320, 1280, scale=0.03, seed=2).to(DEV)
    out, ref = torch.zeros(256, 320, dtype=torch.float16, device=DEV), torch.zeros(256, 320, dtype=torch.float16, device=DEV)
    ops.gemm(A, B, out, tile_hint=19)  # K = 1280 > 768: runs as the heuristic tile
    ops.gemm(A, B, ref, tile_hint=0, split_k=1)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    g, b = torch.ones(1280, device=DEV), torch.zeros(1280, device=DEV)
    with pytest.raises(RuntimeError, match="LayerNorm"):
        ops.gemm(A, B, out, tile_hint=19, ln=(g, b, None, None, 1e-5))
    with pytest.raises(RuntimeError, match="LayerNorm"):
        ops.gemm(A[:, :320], B[:, :320], out, tile_hint=13, ln=(g[:320], b[:320], None, None, 1e-5))
