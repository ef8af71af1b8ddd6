"""Per-kernel parity: each HIP kernel (through the C ABI) vs a plain PyTorch fp32 CPU reference
computed from the SAME f16-rounded inputs.  Tolerances are stated per test."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from view_neti_amd import ops
    return ops


def rnd(*shape, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def relerr(a, b):
    a = a.float().cpu()
    b = b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item(), (a - b).abs().max().item()


ELEM_K = 8.0


def check(name, got, ref, tol, elem_k=ELEM_K):
    """two bounds: the relative Frobenius norm (< tol), and an ELEMENT-WISE one so that a wrong tail row / column of a
    large output cannot hide inside the norm: |got - ref| <= elem_k * tol * (rms(ref) + |ref|) for every element
    (f16 rounding scales with the element, accumulation noise with the tensor's rms; a Gaussian error of sigma =
    tol * rms stays under 6 sigma over 1e7 elements, a dropped or misplaced element is off by ~|ref| itself)."""
    a, b = got.float().cpu(), ref.float().cpu()
    r, m = relerr(a, b)
    rms = b.pow(2).mean().sqrt()
    excess = (a - b).abs() - elem_k * tol * (rms + b.abs())
    worst = excess.max().item()
    print(f"[{name}] rel={r:.3e} maxabs={m:.3e} elem-bound margin={-worst:.3e}")
    assert math.isfinite(r) and r < tol, f"{name}: rel err {r} (max abs {m}) >= {tol}"
    if worst > 0:
        idx = [int(i) for i in torch.unravel_index(excess.argmax(), excess.shape)]
        raise AssertionError(f"{name}: element {idx} off by {(a - b)[tuple(idx)].item():.4e} (ref {b[tuple(idx)].item():.4e}, "
                             f"rms {rms.item():.3e}): beyond the element-wise bound {elem_k}*{tol}*(rms+|ref|)")


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("hint", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 101, 102, 103, 104, 105])
@pytest.mark.parametrize("M,N,K", [(300, 320, 128), (128, 64, 64), (77 * 4, 1280, 768), (1000, 8, 192)])
def test_gemm_plain(hint, M, N, K):
    ops = _ops()
    A = rnd(M, K, seed=1)
    B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=2)
    bias = rnd(N, seed=3, dtype=torch.float32)
    res = rnd(M, N, seed=4)
    ref = A.float() @ B.float().t() + bias
    ref_h = ref.half().float() + res.float()
    out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), resid=res.to(DEV), tile_hint=hint)
    torch.cuda.synchronize()
    check(f"gemm {M}x{N}x{K} hint{hint}", out, ref_h, 2e-3)


def test_gemm_transpose_detect_and_strides():
    """asymmetric B and strided views (ld > cols) — catches swapped operand / layout bugs."""
    ops = _ops()
    M, N, K = 192, 136, 64
    Abig = rnd(M, K + 64, seed=5).to(DEV)
    A = Abig[:, 64:]
    B = torch.zeros(N, K, dtype=torch.float16)
    for n in range(N):
        B[n, (n * 7) % K] = 1.0 + n / 64.0
    outbig = torch.zeros(M, N + 24, dtype=torch.float16, device=DEV)
    out = outbig[:, 8:8 + N]
    ops.gemm(A, B.to(DEV), out, alpha=0.5)
    torch.cuda.synchronize()
    ref = 0.5 * (A.float().cpu() @ B.float().t())
    check("gemm strided/asym", out, ref, 1e-3)
    assert outbig[:, :8].abs().sum().item() == 0 and outbig[:, 8 + N:].abs().sum().item() == 0


def test_gemm_f32_out_rowadd_act_batch():
    ops = _ops()
    M, N, K = 260, 192, 128
    A = rnd(M, K, seed=6)
    B = rnd(N, K, scale=1 / math.sqrt(K), seed=7)
    res = rnd(M, N, seed=8, dtype=torch.float32)
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), out, resid=res.to(DEV))
    torch.cuda.synchronize()
    check("gemm f32out+resid", out, A.float() @ B.float().t() + res, 1e-3)
    # rowadd (time-embedding broadcast) + SiLU epilogue, f16
    rows_per_group = 65
    radd = rnd(M // rows_per_group, N, seed=9)
    out2 = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(A.to(DEV), B.to(DEV), out2, rowadd=radd.to(DEV), rows_per_group=rows_per_group, act=ops.ACT_SILU)
    torch.cuda.synchronize()
    ref = F.silu(A.float() @ B.float().t()).half().float() + radd.float().repeat_interleave(rows_per_group, 0)
    check("gemm rowadd+silu", out2, ref, 2e-3)
    # batched (grid.y) with shared B
    Ab = rnd(3, 100, K, seed=10)
    outb = torch.zeros(3, 100, N, dtype=torch.float16, device=DEV)
    ops.gemm(Ab.to(DEV), B.to(DEV), outb, batch=3, strideA=100 * K, strideB=0, strideC=100 * N, M=100, lda=K, ldc=N)
    torch.cuda.synchronize()
    check("gemm batched", outb, Ab.float() @ B.float().t(), 2e-3)


@pytest.mark.parametrize("hint", [3, 16, 17])
@pytest.mark.parametrize("split", [2, 3, 5, 0])
@pytest.mark.parametrize("f32", [False, True])
def test_gemm_split_k(split, f32, hint):
    """split-K partials + reduce kernel must reproduce the fused epilogue (bias, row-add, residual)."""
    ops = _ops()
    M, N, K = 200, 328, 64 * 24
    A = rnd(M, K, seed=41)
    B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=42)
    bias = rnd(N, seed=43, dtype=torch.float32)
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=DEV)
    if f32:
        res = rnd(M, N, seed=44, dtype=torch.float32)
        out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        ops.gemm(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), resid=res.to(DEV), workspace=ws, split_k=split, tile_hint=hint)
        ref = A.float() @ B.float().t() + bias + res
    else:
        res = rnd(M, N, seed=44)
        radd = rnd(M // 50, N, seed=45)
        out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
        ops.gemm(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), resid=res.to(DEV), rowadd=radd.to(DEV), rows_per_group=50,
                 workspace=ws, split_k=split, tile_hint=hint)
        ref = (A.float() @ B.float().t() + bias).half().float() + radd.float().repeat_interleave(50, 0) + res.float()
    torch.cuda.synchronize()
    check(f"gemm split_k={split} f32={f32}", out, ref, 2e-3)


@pytest.mark.parametrize("hint", [3, 9, 13, 16, 17])
def test_gemm_epilogue_adds_round_like_f32(hint):
    """the fused row-add and residual are 16-bit adds (packed v_pk_add_f16 in the fp16 build): bit for bit
    half(float(half(float(c) + float(rowadd))) + float(resid)) of the plain launch's output c — including operands many
    binades apart and on rounding ties."""
    ops = _ops()
    M, N, K = 256, 320, 128
    A = rnd(M, K, seed=61).to(DEV)
    B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=62).to(DEV)
    g = torch.Generator().manual_seed(63)
    # residual / row-add magnitudes spread over 2^-14 .. 2^6 so that every alignment case of the adder occurs
    res = (torch.randn(M, N, generator=g) * torch.exp2(torch.randint(-14, 7, (M, N), generator=g).float())).to(torch.float16).to(DEV)
    radd = (torch.randn(M // 64, N, generator=g) * torch.exp2(torch.randint(-10, 3, (M // 64, N), generator=g).float())).to(torch.float16).to(DEV)
    plain = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    fused = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(A, B, plain, tile_hint=hint, split_k=1)
    ops.gemm(A, B, fused, resid=res, rowadd=radd, rows_per_group=64, tile_hint=hint, split_k=1)
    torch.cuda.synchronize()
    want = ((plain.float() + radd.float().repeat_interleave(64, 0)).to(torch.float16).float() + res.float()).to(torch.float16)
    assert torch.equal(fused, want)


@pytest.mark.parametrize("hint", [3, 16, 17])
@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("split", [1, 3])
@pytest.mark.parametrize("N", [328, 324])
def test_gemm_gate_and_second_output(act, split, N, hint):
    """epilogue extensions of the CLIP MLP: out = (A@B^T + bias) * act'(pre), out2 = act2(out)
    (transformers CLIPMLP forward/backward, modeling_clip.py) in the fused and the split-K reduce epilogue."""
    ops = _ops()
    M, K = 200, 64 * 6
    A = rnd(M, K, seed=51)
    B = rnd(N, K, scale=1.0 / math.sqrt(K), seed=52)
    bias = rnd(N, seed=53, dtype=torch.float32)
    ldp = 336
    pre = rnd(M, ldp, seed=54, scale=1.5)
    ws = torch.empty(8 * M * N, dtype=torch.float32, device=DEV)
    out = torch.zeros(M, ldp, dtype=torch.float16, device=DEV)[:, :N]
    out2 = torch.zeros(M, ldp, dtype=torch.float16, device=DEV)[:, :N]
    ops.gemm(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), gate=pre.to(DEV)[:, :N], gate_act=act, out2=out2, act2=act,
             workspace=ws, split_k=split, tile_hint=hint)
    torch.cuda.synchronize()
    x = pre[:, :N].float().requires_grad_(True)
    fn = {1: F.silu, 2: lambda t: t * torch.sigmoid(1.702 * t), 3: F.gelu}[act]
    fn(x).sum().backward()
    ref = ((A.float() @ B.float().t() + bias).half().float() * x.grad).half().float()
    check(f"gemm gate act{act} split{split}", out, ref, 2e-3)
    check(f"gemm out2 act{act} split{split}", out2, fn(out.float().cpu()), 1e-3)


@pytest.mark.parametrize("hint", [0, 1, 3, 5, 7, 10, 12, 16, 17])
@pytest.mark.parametrize("M,C", [(300, 64), (1024, 320)])
def test_gemm_geglu_epilogues(hint, M, C):
    """diffusers FeedForward GEGLU (h, g = proj(x).chunk(2); h * gelu(g)) fused into the GEMM epilogues
    (vneti_gemm_desc.geglu): forward = the interleaved pre-activation + the gated output from ONE launch; backward =
    the ff.net.2 dgrad GEMM writing d_h | d_g without storing d(h*gelu(g)).  Reference: torch autograd on the same
    f16-rounded tensors, un-interleaved."""
    from view_neti_amd import packing
    ops = _ops()
    F4 = 4 * C
    x = rnd(M, C, seed=61)
    W = rnd(2 * F4, C, scale=1.0 / math.sqrt(C), seed=62)
    b = rnd(2 * F4, seed=63, dtype=torch.float32)
    idx = packing.geglu_interleave_index(2 * F4)
    p = torch.zeros(M, 2 * F4, dtype=torch.float16, device=DEV)
    gg = torch.zeros(M, F4, dtype=torch.float16, device=DEV)
    ops.gemm(x.to(DEV), packing.geglu_interleave(W).to(DEV), p, bias=packing.geglu_interleave(b).to(DEV), out2=gg, geglu=1,
             split_k=1, tile_hint=hint)
    torch.cuda.synchronize()
    pre = (x.float() @ W.float().t() + b).half().float()            # [h | g], reference order
    check(f"geglu fwd pre-activation (interleaved) hint{hint}", p, pre[:, idx], 2e-3)
    pg = p.float().cpu()
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(2 * F4)
    pu = pg[:, inv]                                                  # the GPU's own pre-activation, un-interleaved
    check(f"geglu fwd gate hint{hint}", gg, pu[:, :F4] * F.gelu(pu[:, F4:]), 1e-3)
    # backward: d = dy @ W2 (ff.net.2 dgrad), then through h * gelu(g)
    dy = rnd(M, C, seed=64)
    W2 = rnd(C, F4, scale=1.0 / math.sqrt(F4), seed=65)             # ff.net.2.weight [C][4C]
    dp = torch.zeros(M, 2 * F4, dtype=torch.float16, device=DEV)
    ops.gemm(dy.to(DEV), W2.t().contiguous().to(DEV), dp, gate=p, gate_act=ops.ACT_GELU, geglu=2, split_k=1, tile_hint=hint)
    torch.cuda.synchronize()
    d = (dy.float() @ W2.float()).half().float()
    hg = pu.clone().requires_grad_(True)
    (hg[:, :F4] * F.gelu(hg[:, F4:]) * d).sum().backward()
    check(f"geglu bwd hint{hint}", dp, hg.grad[:, idx], 2e-3)
    with pytest.raises(RuntimeError, match="split_k"):
        ops.gemm(x.to(DEV), packing.geglu_interleave(W).to(DEV), p, out2=gg, geglu=1, split_k=2, tile_hint=3,
                 workspace=torch.empty(2 * M * 2 * F4, dtype=torch.float32, device=DEV))


@pytest.mark.parametrize("hint", [0, 1, 2, 3, 5, 7, 9, 14, 16, 17])
@pytest.mark.parametrize("Bn,HW,C", [(4, 64, 320), (2, 256, 128), (3, 576, 640), (2, 4096, 32 * 4)])
def test_gemm_groupnorm_sums_and_fused_apply(hint, Bn, HW, C):
    """GroupNorm statistics accumulated by the GEMM epilogue (rows = Bn images of HW pixels, 32 groups) and the
    one-launch GroupNorm that consumes them, against torch.nn.functional.group_norm of the SAME f16 tensor
    (ResnetBlock2D.norm1/norm2 + SiLU, diffusers resnet.py)."""
    ops = _ops()
    G, S, K = 32, (8 if hint != 3 else 19), 128  # (19 slots: the consumer's batches beyond the first eight, ragged)
    M = Bn * HW
    A = rnd(M, K, seed=61)
    B = rnd(C, K, scale=1.0 / math.sqrt(K), seed=62)
    bias = rnd(C, seed=63, dtype=torch.float32) * 2
    res = rnd(M, C, seed=64)
    out = torch.zeros(M, C + 8, dtype=torch.float16, device=DEV)[:, :C]
    sums = torch.zeros(Bn, S, G, 4, dtype=torch.int64, device=DEV)  # fixed-point slot sums (csrc/common.h vn_fx_*)
    ops.gemm(A.to(DEV), B.to(DEV), out, bias=bias.to(DEV), resid=res.to(DEV), tile_hint=hint, split_k=1,
             gn_sums=sums, gn_hw=HW, gn_groups=G, gn_slots=S)
    torch.cuda.synchronize()
    x = out.float().cpu().reshape(Bn, HW, G, C // G)
    ref_s = x.sum((1, 3))
    ref_q = (x * x).sum((1, 3))
    got = ops.gn_sums_decode(sums.sum(1))
    check(f"gn sums hint{hint}", got[..., 0], ref_s, 1e-3)
    check(f"gn sumsq hint{hint}", got[..., 1], ref_q, 1e-5)
    gamma = rnd(C, seed=65, dtype=torch.float32) * 0.1 + 1
    beta = rnd(C, seed=66, dtype=torch.float32) * 0.1
    y = torch.zeros(M, C, dtype=torch.float16, device=DEV)
    mean = torch.zeros(Bn * G, dtype=torch.float32, device=DEV)
    rstd = torch.zeros(Bn * G, dtype=torch.float32, device=DEV)
    ops.groupnorm_fwd_sums(out, y, gamma.to(DEV), beta.to(DEV), sums, S, mean, rstd, Bn, HW, C, G, 1e-5, True)
    torch.cuda.synchronize()
    xn = out.float().cpu().reshape(Bn, HW, C).permute(0, 2, 1)
    ref = F.silu(F.group_norm(xn, G, gamma, beta, 1e-5)).permute(0, 2, 1).reshape(M, C)
    check(f"gn fused apply hint{hint}", y, ref, 2e-3)
    xg = xn.reshape(Bn, G, -1)
    check("gn mean", mean.cpu().reshape(Bn, G), xg.mean(-1), 1e-4)
    check("gn rstd", rstd.cpu().reshape(Bn, G), (xg.var(-1, unbiased=False) + 1e-5).rsqrt(), 1e-4)


# ------------------------------------------------------------------------------------------ conv
def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("korder", [0, 1], ids=["tap-major", "chunk-major"])
@pytest.mark.parametrize("case", ["s1", "s2p1", "s2vae", "ups"])
@pytest.mark.parametrize("hint", [0, 1, 3, 4, 5, 6, 7, 8, 9, 13, 15, 16, 17, 18, 101, 103])
def test_conv3x3_fwd(case, hint, korder):
    """both K orders of the implicit GEMM (vneti_gemm_desc.conv_korder): (tap, channel) and (64-channel chunk, tap,
    channel); two chunks so that the orders really differ"""
    ops = _ops()
    from view_neti_amd import packing
    Bn, Ci, Co, H, W = 2, 128, 128, 12, 20
    x = rnd(Bn, Ci, H, W, seed=11)
    w = rnd(Co, Ci, 3, 3, scale=1 / math.sqrt(9 * Ci), seed=12)
    bias = rnd(Co, seed=13, dtype=torch.float32)
    xf, wf = x.float(), w.float()
    if case == "s1":
        ref = F.conv2d(xf, wf, bias, padding=1)
        conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci)
    elif case == "s2p1":
        ref = F.conv2d(xf, wf, bias, stride=2, padding=1)
        conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H // 2, Wo=W // 2, stride=2, pad_t=1, pad_l=1, ups=0, ldx=Ci)
    elif case == "s2vae":
        ref = F.conv2d(F.pad(xf, (0, 1, 0, 1)), wf, bias, stride=2, padding=0)
        conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H // 2, Wo=W // 2, stride=2, pad_t=0, pad_l=0, ups=0, ldx=Ci)
    else:
        ref = F.conv2d(F.interpolate(xf, scale_factor=2.0, mode="nearest"), wf, bias, padding=1)
        conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=2 * H, Wo=2 * W, stride=1, pad_t=1, pad_l=1, ups=1, ldx=Ci)
    Ho, Wo = ref.shape[2], ref.shape[3]
    out = torch.zeros(Bn * Ho * Wo, Co, dtype=torch.float16, device=DEV)
    conv["korder"] = korder
    ops.gemm(_nhwc(x).to(DEV), packing.conv3x3_fwd(w, cm=bool(korder)).to(DEV), out, bias=bias.to(DEV), conv=conv,
             M=Bn * Ho * Wo, tile_hint=hint)
    torch.cuda.synchronize()
    check(f"conv {case} hint{hint} korder{korder}", out.view(Bn, Ho, Wo, Co), _nhwc(ref), 2e-3)


@pytest.mark.parametrize("shape", [(2, 128, 128, 32, 48), (1, 192, 320, 16, 32), (2, 256, 96, 32, 16), (3, 64, 128, 16, 16)],
                         ids=["2chunks", "3chunks-Ntail", "4chunks-N96", "1chunk"])
@pytest.mark.parametrize("epi", ["bias", "resid+rowadd+gn", "silu"])
def test_conv3x3_halo_tile(shape, epi):
    """tile_hint 18 (16 x 16-pixel blocks, the 18 x 18 input patch resident in LDS for all nine taps) against F.conv2d and,
    bit for bit, against the row-major 8-phase tile 17 — with the fused epilogue operands of the UNet's resnet convs (time
    embedding row-add, residual, GroupNorm sums of the output) addressed through the tile's pixel order"""
    ops = _ops()
    from view_neti_amd import packing
    Bn, Ci, Co, H, W = shape
    x = rnd(Bn, Ci, H, W, seed=21)
    w = rnd(Co, Ci, 3, 3, scale=1 / math.sqrt(9 * Ci), seed=22)
    bias = rnd(Co, seed=23, dtype=torch.float32)
    M = Bn * H * W
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    kw = dict(bias=bias.to(DEV))
    G = Co // 8  # (the fused statistics need at least 8 channels per group)
    if epi == "resid+rowadd+gn":
        res = rnd(M, Co, seed=24)
        temb = rnd(Bn, Co, seed=25)
        ref = ref + res.float().view(Bn, H, W, Co).permute(0, 3, 1, 2) + temb.float()[:, :, None, None]
        kw.update(resid=res.to(DEV), rowadd=temb.to(DEV), rows_per_group=H * W)
    elif epi == "silu":
        ref = F.silu(ref)
        kw.update(act=1)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
    outs, sums = {}, {}
    for hint in (17, 18):
        out = torch.zeros(M, Co, dtype=torch.float16, device=DEV)
        k2 = dict(kw)
        if epi == "resid+rowadd+gn":
            sums[hint] = torch.zeros(Bn, 4, G, 4, dtype=torch.int64, device=DEV)
            k2.update(gn_sums=sums[hint], gn_hw=H * W, gn_groups=G, gn_slots=4)
        ops.gemm(_nhwc(x).to(DEV), packing.conv3x3_fwd(w, cm=True).to(DEV), out, conv=conv, M=M, tile_hint=hint, split_k=1, **k2)
        torch.cuda.synchronize()
        outs[hint] = out
    check(f"conv halo {shape} {epi}", outs[18].view(Bn, H, W, Co), _nhwc(ref), 4e-3)
    assert torch.equal(outs[17], outs[18]), "tile 18 differs from tile 17"
    if sums:
        # (the f32 per-thread partials group the pixels differently in the two orders: equal to f32 rounding, not bit for bit)
        got, got17 = ops.gn_sums_decode(sums[18].sum(1)), ops.gn_sums_decode(sums[17].sum(1))
        assert (got - got17).abs().max() <= 1e-6 * got17.abs().max(), "GroupNorm sums differ between the pixel orders"
        xg = outs[18].double().cpu().view(Bn, H * W, G, Co // G)
        check("halo gn sum", got[..., 0].float(), xg.sum((1, 3)).float(), 1e-4)
        check("halo gn sumsq", got[..., 1].float(), (xg * xg).sum((1, 3)).float(), 1e-4)


@pytest.mark.parametrize("shape,split", [((1, 192, 320, 16, 32), 2), ((2, 256, 96, 32, 16), 3), ((1, 640, 128, 16, 16), 4)],
                         ids=["3chunks/2", "4chunks/3", "10chunks/4"])
def test_conv3x3_halo_tile_split_k(shape, split):
    """tile 18 with split-K: the splits are rounded to whole 64-channel chunks (3 chunks / 2 -> 2 + 1, 4 / 3 -> 2 + 2,
    10 / 4 -> 3 + 3 + 3 + 1); f32 partials + the reduce kernel's epilogue (bias, residual)"""
    ops = _ops()
    from view_neti_amd import packing
    Bn, Ci, Co, H, W = shape
    x = rnd(Bn, Ci, H, W, seed=41)
    w = rnd(Co, Ci, 3, 3, scale=1 / math.sqrt(9 * Ci), seed=42)
    bias = rnd(Co, seed=43, dtype=torch.float32)
    M = Bn * H * W
    res = rnd(M, Co, seed=44)
    ref = F.conv2d(x.float(), w.float(), bias, padding=1) + res.float().view(Bn, H, W, Co).permute(0, 3, 1, 2)
    conv = dict(mode=1, Hi=H, Wi=W, Ci=Ci, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Ci, korder=1)
    ws = torch.empty(split * M * Co, dtype=torch.float32, device=DEV)
    out = torch.zeros(M, Co, dtype=torch.float16, device=DEV)
    ops.gemm(_nhwc(x).to(DEV), packing.conv3x3_fwd(w, cm=True).to(DEV), out, bias=bias.to(DEV), resid=res.to(DEV), conv=conv, M=M,
             tile_hint=18, split_k=split, workspace=ws)
    torch.cuda.synchronize()
    check(f"conv halo split {shape} /{split}", out.view(Bn, H, W, Co), _nhwc(ref), 3e-3)


@pytest.mark.parametrize("gate", [False, True], ids=["plain", "silu-gate"])
@pytest.mark.parametrize("shape", [(2, 128, 64, 16, 32), (1, 192, 320, 32, 16)], ids=["2chunks", "3chunks-Ntail"])
def test_conv3x3_dgrad_halo_tile(shape, gate):
    """tile_hint 18 on the stride-1 transposed gather (the input gradient of the UNet's 3x3 convolutions; the tap (dy, dx)
    reads patch pixel (y + 2 - dy, x + 2 - dx)): against autograd and bit for bit against tile 17; with the fused
    activation-gradient gate of the GroupNorm-SiLU backward"""
    ops = _ops()
    from view_neti_amd import packing
    Bn, Co, Ci, H, W = shape  # dy has Co channels, dx has Ci
    x = rnd(Bn, Ci, H, W, seed=31).float().requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, scale=1 / math.sqrt(9 * Ci), seed=32)
    y = F.conv2d(x, w.float(), None, padding=1)
    dy = rnd(*y.shape, seed=33)
    y.backward(dy.float())
    ref = x.grad
    M = Bn * H * W
    kw = {}
    if gate:
        pre = rnd(M, Ci, seed=34)
        s = torch.sigmoid(pre.float())
        ref = ref * (s * (1 + pre.float() * (1 - s))).view(Bn, H, W, Ci).permute(0, 3, 1, 2)
        kw.update(gate=pre.to(DEV), gate_act=1)
    conv = dict(mode=2, Hi=H, Wi=W, Ci=Co, Ho=H, Wo=W, stride=1, pad_t=1, pad_l=1, ups=0, ldx=Co, korder=1)
    outs = {}
    for hint in (17, 18):
        dx = torch.zeros(M, Ci, dtype=torch.float16, device=DEV)
        ops.gemm(_nhwc(dy).to(DEV), packing.conv3x3_dgrad(w, cm=True).to(DEV), dx, conv=conv, M=M, split_k=1, tile_hint=hint, **kw)
        torch.cuda.synchronize()
        outs[hint] = dx
    check(f"conv dgrad halo {shape} gate={gate}", outs[18].view(Bn, H, W, Ci), _nhwc(ref), 3e-3)
    assert torch.equal(outs[17], outs[18]), "tile 18 differs from tile 17"


@pytest.mark.parametrize("hint", [3, 16, 17, 18])
@pytest.mark.parametrize("korder", [0, 1], ids=["tap-major", "chunk-major"])
@pytest.mark.parametrize("split", [1, 3])
@pytest.mark.parametrize("stride,vae", [(1, False), (2, False), (2, True)])
def test_conv3x3_dgrad(stride, vae, split, korder, hint):
    ops = _ops()
    from view_neti_amd import packing
    Bn, Ci, Co, H, W = 2, 64, 128, 12, 20
    x = rnd(Bn, Ci, H, W, seed=14).float().requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, scale=1 / math.sqrt(9 * Ci), seed=15)
    if vae:
        y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w.float(), None, stride=2, padding=0)
        pt = 0
    else:
        y = F.conv2d(x, w.float(), None, stride=stride, padding=1)
        pt = 1
    dy = rnd(*y.shape, seed=16)
    y.backward(dy.float())
    Ho, Wo = y.shape[2], y.shape[3]
    dx = torch.zeros(Bn * H * W, Ci, dtype=torch.float16, device=DEV)
    conv = dict(mode=2, Hi=Ho, Wi=Wo, Ci=Co, Ho=H, Wo=W, stride=stride, pad_t=pt, pad_l=pt, ups=0, ldx=Co, korder=korder)
    ws = torch.empty(4 * Bn * H * W * Ci, dtype=torch.float32, device=DEV)
    ops.gemm(_nhwc(dy).to(DEV), packing.conv3x3_dgrad(w, cm=bool(korder)).to(DEV), dx, conv=conv, M=Bn * H * W,
             workspace=ws, split_k=split, tile_hint=hint)
    torch.cuda.synchronize()
    check(f"conv dgrad s{stride} vae={vae} split{split} korder{korder}", dx.view(Bn, H, W, Ci), _nhwc(x.grad), 2e-3)


def test_im2col_small_and_conv_in():
    ops = _ops()
    from view_neti_amd import packing
    Bn, Ci, Co, H, W = 2, 4, 64, 16, 16
    x = rnd(Bn, Ci, H, W, seed=17, dtype=torch.float32)
    w = rnd(Co, Ci, 3, 3, scale=0.2, seed=18)
    col = torch.zeros(Bn * H * W, 64, dtype=torch.float16, device=DEV)
    xd = x.to(DEV)
    ops.im2col3x3_small(xd, col, Bn, Ci, H, W, H, W, 1, 1, 1, xd.stride())
    wp = packing.pad_rows(packing.conv3x3_fwd(w), 64).to(DEV)
    out = torch.zeros(Bn * H * W, Co, dtype=torch.float16, device=DEV)
    ops.gemm(col, wp, out)
    torch.cuda.synchronize()
    ref = F.conv2d(x.half().float(), w.float(), None, padding=1)
    check("conv_in via im2col", out.view(Bn, H, W, Co), _nhwc(ref), 2e-3)


@pytest.mark.parametrize("Ci,Co,H,W,f32,gn", [(3, 128, 32, 48, True, True), (3, 256, 16, 16, False, True),
                                              (1, 128, 9, 21, True, False), (2, 128, 40, 24, False, False)])
def test_conv3x3_in_direct(Ci, Co, H, W, f32, gn):
    """AutoencoderKL encoder.conv_in straight from the (strided) pixels, with the GroupNorm sums of the stored values"""
    ops = _ops()
    from view_neti_amd import packing
    Bn, G, S = 3, 32, 8
    x = rnd(Bn, Ci + 1, H + 2, W + 3, seed=21, dtype=torch.float32)
    xd = (x if f32 else x.half()).to(DEV)[:, :Ci, 1:H + 1, 2:W + 2]     # a strided view: channel, row and batch strides
    w = rnd(Co, Ci, 3, 3, scale=0.3, seed=22)
    bias = rnd(Co, seed=23, dtype=torch.float32)
    out = torch.zeros(Bn * H * W, Co + 8, dtype=torch.float16, device=DEV)[:, :Co]
    sums = torch.zeros(Bn, S, G, 4, dtype=torch.int64, device=DEV) if gn else None
    ops.conv3x3_in(xd, packing.conv_in_direct(w).to(DEV), bias.to(DEV), out, Bn, Ci, H, W, xd.stride(),
                   gn_sums=sums, gn_hw=H * W if gn else 0, gn_groups=G if gn else 0, gn_slots=S if gn else 0)
    torch.cuda.synchronize()
    xr = x[:, :Ci, 1:H + 1, 2:W + 2]
    ref = F.conv2d(xr.half().float(), w.float(), bias, padding=1)
    check(f"conv_in direct C{Ci} Co{Co}", out.reshape(Bn, H, W, Co), _nhwc(ref), 2e-3)
    if gn:
        xo = out.float().cpu().reshape(Bn, H * W, G, Co // G)
        got = ops.gn_sums_decode(sums.sum(1))
        check("conv_in gn sums", got[..., 0], xo.sum((1, 3)), 1e-3)
        check("conv_in gn sumsq", got[..., 1], (xo * xo).sum((1, 3)), 1e-5)


# ------------------------------------------------------------------------------------------ norms
# the last twelve complete SURVEY App. B's 14 (C, HW) GroupNorm shapes of the SD-1.5 UNet at 64x64 latents
@pytest.mark.parametrize("Cc,HW", [(320, 256), (128, 1024), (2560, 64), (960, 100), (320, 4096), (128, 40000),
                                   (640, 4096), (960, 4096), (320, 1024), (640, 1024), (960, 1024), (1280, 1024),
                                   (1920, 1024), (640, 256), (1280, 256), (1920, 256), (2560, 256), (1280, 64)])
@pytest.mark.parametrize("silu", [True, False])
@pytest.mark.parametrize("three_launch", [False, True])
def test_groupnorm_fwd_bwd(Cc, HW, silu, three_launch, monkeypatch):
    """both forms: the one-launch kernel small (sample, group) slices dispatch to, and the stats/finalize/apply form"""
    ops = _ops()
    if three_launch:
        monkeypatch.setenv("VNETI_GN_NO_SMALL", "1")
    Bn, G, eps = 2, 32, 1e-5
    x = (rnd(Bn, HW, Cc, seed=19).float() * 1.5 + 0.3).half()
    gamma = 1 + 0.1 * rnd(Cc, seed=20, dtype=torch.float32)
    beta = 0.1 * rnd(Cc, seed=21, dtype=torch.float32)
    dy = rnd(Bn, HW, Cc, seed=22)
    xr = x.float().permute(0, 2, 1).requires_grad_(True)  # [B, C, HW]
    yr = F.group_norm(xr, G, gamma, beta, eps)
    if silu:
        yr = F.silu(yr)
    yr.backward(dy.float().permute(0, 2, 1))
    xd = x.to(DEV).view(Bn * HW, Cc)
    y = torch.zeros_like(xd)
    mean = torch.zeros(Bn * G, dtype=torch.float32, device=DEV)
    rstd = torch.zeros_like(mean)
    ws = torch.zeros(ops.groupnorm_ws_floats(Bn, HW, Cc, G), dtype=torch.float32, device=DEV)
    ops.groupnorm_fwd(xd, y, gamma.to(DEV), beta.to(DEV), mean, rstd, ws, Bn, HW, Cc, G, eps, silu)
    dx = torch.zeros_like(xd)
    acc = rnd(Bn * HW, Cc, seed=23).to(DEV)
    ops.groupnorm_bwd(dy.to(DEV).view(Bn * HW, Cc), xd, gamma.to(DEV), beta.to(DEV), mean, rstd, dx, ws, Bn, HW,
                      Cc, G, silu, accum=acc)
    torch.cuda.synchronize()
    check(f"gn fwd C{Cc} HW{HW} silu{silu}", y.view(Bn, HW, Cc), yr.detach().permute(0, 2, 1), 2e-3)
    check(f"gn bwd C{Cc} HW{HW} silu{silu}", dx.view(Bn, HW, Cc).float().cpu() - acc.view(Bn, HW, Cc).float().cpu(),
          xr.grad.permute(0, 2, 1), 4e-3)
    # the two-launch form (statistics into caller-zeroed slot sums, finalize folded into the apply kernel)
    S = 8
    y2, dx2 = torch.zeros_like(xd), torch.zeros_like(xd)
    mean2, rstd2 = torch.zeros_like(mean), torch.zeros_like(mean)
    fs = torch.zeros(Bn, S, G, 4, dtype=torch.int64, device=DEV)
    bs = torch.zeros(Bn, S, G, 4, dtype=torch.int64, device=DEV)
    ops.groupnorm_fwd_2l(xd, y2, gamma.to(DEV), beta.to(DEV), fs, S, mean2, rstd2, Bn, HW, Cc, G, eps, silu)
    ops.groupnorm_bwd_2l(dy.to(DEV).view(Bn * HW, Cc), xd, gamma.to(DEV), beta.to(DEV), mean2, rstd2, dx2, bs, S, ws, Bn,
                         HW, Cc, G, silu, accum=acc)
    torch.cuda.synchronize()
    check(f"gn 2l fwd C{Cc} HW{HW} silu{silu}", y2.view(Bn, HW, Cc), yr.detach().permute(0, 2, 1), 2e-3)
    check(f"gn 2l bwd C{Cc} HW{HW} silu{silu}", dx2.view(Bn, HW, Cc).float().cpu() - acc.view(Bn, HW, Cc).float().cpu(),
          xr.grad.permute(0, 2, 1), 4e-3)
    check("gn 2l mean", mean2, mean, 1e-5)
    check("gn 2l rstd", rstd2, rstd, 1e-5)


@pytest.mark.parametrize("Cc", [320, 768, 1280])
@pytest.mark.parametrize("f32", [False, True])
def test_layernorm_fwd_bwd(Cc, f32):
    ops = _ops()
    rows = 77
    dt = torch.float32 if f32 else torch.float16
    x = (rnd(rows, Cc, seed=24, dtype=torch.float32) * 2 + 0.5).to(dt)
    gamma = 1 + 0.1 * rnd(Cc, seed=25, dtype=torch.float32)
    beta = 0.1 * rnd(Cc, seed=26, dtype=torch.float32)
    dy = rnd(rows, Cc, seed=27)
    xr = x.float().requires_grad_(True)
    yr = F.layer_norm(xr, (Cc,), gamma, beta, 1e-5)
    yr.backward(dy.float())
    xd = x.to(DEV)
    y = torch.zeros(rows, Cc, dtype=torch.float16, device=DEV)
    mean = torch.zeros(rows, dtype=torch.float32, device=DEV)
    rstd = torch.zeros_like(mean)
    ops.layernorm_fwd(xd, y, gamma.to(DEV), beta.to(DEV), mean, rstd, 1e-5)
    dx = torch.zeros(rows, Cc, dtype=dt, device=DEV)
    acc = rnd(rows, Cc, seed=28, dtype=torch.float32).to(dt).to(DEV)
    ops.layernorm_bwd(dy.to(DEV), xd, gamma.to(DEV), mean, rstd, dx, accum=acc)
    torch.cuda.synchronize()
    check(f"ln fwd C{Cc} f32={f32}", y, yr.detach(), 2e-3)
    check(f"ln bwd C{Cc} f32={f32}", dx.float().cpu() - acc.float().cpu(), xr.grad, 4e-3)


def test_transpose_and_softmax():
    ops = _ops()
    Bn, rows, cols, ldo = 3, 77, 320, 80
    x = rnd(Bn, rows, cols, seed=29).to(DEV)
    out = torch.full((Bn, cols, ldo), 7.0, dtype=torch.float16, device=DEV)
    ops.transpose(x, out, rows, cols, Bn, cols, rows * cols, ldo, cols * ldo)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :, :rows], x.transpose(1, 2))
    assert out[:, :, rows:].abs().sum().item() == 0
    s = rnd(50, 4096, scale=3.0, seed=30).to(DEV)
    ref = torch.softmax(s.float().cpu(), -1)
    ops.softmax_rows(s, 50, 4096)
    torch.cuda.synchronize()
    check("softmax rows", s, ref, 2e-3)


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, H, D, scale, causal):
    Bn, Nq, _ = q.shape
    Nk = k.shape[1]
    qh = q.view(Bn, Nq, H, D).transpose(1, 2)
    kh = k.view(Bn, Nk, H, D).transpose(1, 2)
    vh = v.view(Bn, Nk, H, D).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        mask = torch.ones(Nq, Nk, dtype=torch.bool).tril()
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ vh).transpose(1, 2).reshape(Bn, Nq, H * D)
    lse = torch.logsumexp(s, -1)
    return o, lse


@pytest.mark.parametrize("D,H,Nq,Nk,causal", [
    (40, 8, 200, 200, False), (40, 2, 300, 77, False), (64, 12, 77, 77, True), (80, 8, 128, 77, False),
    (160, 8, 256, 256, False), (160, 2, 64, 77, False), (64, 3, 200, 200, True), (80, 2, 520, 520, False),
    (40, 8, 1024, 77, False), (80, 4, 515, 77, False),
    # the step's own shapes (one sample): the dominant 64x64-latent self- and cross-attention, the
    # 32x32 and 16x16 levels
    (40, 8, 4096, 4096, False), (40, 8, 4096, 77, False), (80, 8, 1024, 1024, False), (80, 8, 1024, 77, False),
    (160, 8, 256, 77, False), (160, 8, 64, 64, False),
])
def test_attention_fwd_bwd(D, H, Nq, Nk, causal):
    ops = _ops()
    if ops._default_ws is None:  # enables the q-split dK/dV path for short key sides (cross-attention)
        ops.set_default_gemm_workspace(torch.empty(16 * 2 ** 20, dtype=torch.float32, device=DEV))
    Bn = 1 if Nq >= 4096 else 2
    Cc = H * D
    scale = D ** -0.5
    q = rnd(Bn, Nq, Cc, seed=31)
    k = rnd(Bn, Nk, Cc, seed=32)
    v = rnd(Bn, Nk, Cc, seed=33)
    do = rnd(Bn, Nq, Cc, seed=34)
    # spike one key against one query to exercise the online-softmax rescale path
    k[0, Nk - 3] = (q[0, min(5, Nq - 1)].float() * 3).half()
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    o_ref, lse_ref = _attn_ref(qr, kr, vr, H, D, scale, causal)
    o_ref.backward(do.float())

    qd, kd, vd, dod = (t.to(DEV).view(-1, Cc) for t in (q, k, v, do))
    o = torch.zeros(Bn * Nq, Cc, dtype=torch.float16, device=DEV)
    lse = torch.zeros(Bn, H, Nq, dtype=torch.float32, device=DEV)
    ops.attn_fwd(qd, kd, vd, o, lse, Bn, H, Nq, Nk, D, scale, causal)
    torch.cuda.synchronize()
    tag = f"D{D} H{H} Nq{Nq} Nk{Nk} c{int(causal)}"
    check(f"attn fwd {tag}", o.view(Bn, Nq, Cc), o_ref.detach(), 3e-3)
    check(f"attn lse {tag}", lse, lse_ref.detach(), 1e-3)

    delta = torch.zeros(Bn, H, Nq, dtype=torch.float32, device=DEV)
    ops.attn_bwd_delta(dod, o, delta, Bn, H, Nq, D)
    dq = torch.zeros(Bn * Nq, Cc, dtype=torch.float16, device=DEV)
    dk = torch.zeros(Bn * Nk, Cc, dtype=torch.float16, device=DEV)
    dv = torch.zeros(Bn * Nk, Cc, dtype=torch.float16, device=DEV)
    ops.attn_bwd_dq(qd, kd, vd, dod, lse, delta, dq, Bn, H, Nq, Nk, D, scale, causal)
    # fused variant: delta computed inside the dQ kernel and published for dK/dV
    delta2 = torch.full_like(delta, float("nan"))
    dq2 = torch.zeros_like(dq)
    ops.attn_bwd_dq(qd, kd, vd, dod, lse, delta2, dq2, Bn, H, Nq, Nk, D, scale, causal, O=o)
    torch.cuda.synchronize()
    check(f"attn delta(in-kernel) {tag}", delta2, delta, 1e-4)
    check(f"attn dq(fused delta) {tag}", dq2, dq, 1e-3)
    ops.attn_bwd_dkv(qd, kd, vd, dod, lse, delta2, dk, dv, Bn, H, Nq, Nk, D, scale, causal)
    torch.cuda.synchronize()
    check(f"attn dq {tag}", dq.view(Bn, Nq, Cc), qr.grad, 6e-3)
    check(f"attn dk {tag}", dk.view(Bn, Nk, Cc), kr.grad, 6e-3)
    check(f"attn dv {tag}", dv.view(Bn, Nk, Cc), vr.grad, 6e-3)


@pytest.mark.parametrize("H,N,causal", [(12, 77, True), (16, 77, True), (3, 96, True), (2, 33, True), (4, 77, False), (5, 64, False)])
def test_attention_bwd_small_fused(H, N, causal):
    """vneti_attn_bwd_small (the CLIP text encoder's 77-token, 64-wide-head attention: dQ, dK, dV of a (sequence, head) in one
    launch) against autograd and, bit for bit, against the dQ + dK/dV kernel pair it replaces; strided q / k / v views
    of one fused qkv buffer as the text engine passes them"""
    ops = _ops()
    D, Bn = 64, 3
    Cc = H * D
    scale = D ** -0.5
    qkv = rnd(Bn * N, 3 * Cc, seed=61)
    do = rnd(Bn, N, Cc, seed=62)
    q, k, v = (qkv[:, i * Cc:(i + 1) * Cc].reshape(Bn, N, Cc) for i in range(3))
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, H, D, scale, causal)
    o_ref.backward(do.float())
    qkv_d = qkv.to(DEV)
    qd, kd, vd = qkv_d[:, :Cc], qkv_d[:, Cc:2 * Cc], qkv_d[:, 2 * Cc:]
    dod = do.to(DEV).view(-1, Cc)
    o = torch.zeros(Bn * N, Cc, dtype=torch.float16, device=DEV)
    lse = torch.zeros(Bn, H, N, dtype=torch.float32, device=DEV)
    ops.attn_fwd(qd, kd, vd, o, lse, Bn, H, N, N, D, scale, causal)
    delta = torch.zeros(Bn, H, N, dtype=torch.float32, device=DEV)
    dqkv = [torch.full((Bn * N, 3 * Cc), float("nan"), dtype=torch.float16, device=DEV) for _ in range(2)]
    dq, dk, dv = dqkv[0][:, :Cc], dqkv[0][:, Cc:2 * Cc], dqkv[0][:, 2 * Cc:]
    ops.attn_bwd_dq(qd, kd, vd, dod, lse, delta, dq, Bn, H, N, N, D, scale, causal, O=o)
    ops.attn_bwd_dkv(qd, kd, vd, dod, lse, delta, dk, dv, Bn, H, N, N, D, scale, causal)
    assert ops.attn_bwd_small_ok(N, D)
    dq2, dk2, dv2 = dqkv[1][:, :Cc], dqkv[1][:, Cc:2 * Cc], dqkv[1][:, 2 * Cc:]
    ops.attn_bwd_small(qd, kd, vd, dod, o, lse, dq2, dk2, dv2, Bn, H, N, D, scale, causal)
    torch.cuda.synchronize()
    tag = f"H{H} N{N} c{int(causal)}"
    check(f"attn small dq {tag}", dq2.reshape(Bn, N, Cc), qr.grad, 6e-3)
    check(f"attn small dk {tag}", dk2.reshape(Bn, N, Cc), kr.grad, 6e-3)
    check(f"attn small dv {tag}", dv2.reshape(Bn, N, Cc), vr.grad, 6e-3)
    assert torch.equal(dqkv[0], dqkv[1]), "fused small backward differs from the dQ + dK/dV pair"


def test_vae_single_head_attention_d512():
    """the VAE mid block's attention (one head of dim C = 512 over N = h*w tokens) as the engine issues it: batched
    q.k^T GEMM with alpha = C^-1/2 into an [N, ld] score matrix, softmax_rows in place, v transposed, P.v GEMM — vs
    torch (diffusers 0.14 AttentionBlock; engine/vae.py::_mid_attention)."""
    ops = _ops()
    Bn, N, Cc = 2, 1024, 512  # N = 4096 at 512^2 in the step; 1024 keeps the CPU reference quick, same code path
    ldn = (N + 63) // 64 * 64
    qkv = rnd(Bn * N, 3 * Cc, seed=51, scale=0.7)
    q, k, v = (qkv[:, i * Cc:(i + 1) * Cc].float().view(Bn, N, Cc) for i in range(3))
    ref = torch.softmax(q @ k.transpose(1, 2) * Cc ** -0.5, -1) @ v
    d = qkv.to(DEV)
    qd, kd, vd = d[:, :Cc], d[:, Cc:2 * Cc], d[:, 2 * Cc:]
    scores = torch.zeros(Bn, N, ldn, dtype=torch.float16, device=DEV)
    ops.gemm(qd, kd, scores, alpha=Cc ** -0.5, batch=Bn, strideA=N * 3 * Cc, strideB=N * 3 * Cc, strideC=N * ldn, M=N, N=N,
             K=Cc, lda=3 * Cc, ldc=ldn)
    ops.softmax_rows(scores.view(Bn * N, ldn), Bn * N, N)
    vt = torch.zeros(Bn, Cc, ldn, dtype=torch.float16, device=DEV)
    ops.transpose(vd, vt, N, Cc, Bn, 3 * Cc, N * 3 * Cc, ldn, Cc * ldn)
    o = torch.zeros(Bn * N, Cc, dtype=torch.float16, device=DEV)
    ops.gemm(scores, vt, o, batch=Bn, strideA=N * ldn, strideB=Cc * ldn, strideC=N * Cc, M=N, N=Cc, K=ldn, lda=ldn, ldc=Cc)
    torch.cuda.synchronize()
    check("vae attention d512 probabilities", scores[:, :, :N], torch.softmax(q @ k.transpose(1, 2) * Cc ** -0.5, -1), 3e-3)
    check("vae attention d512 output", o.view(Bn, N, Cc), ref, 3e-3)


def test_transpose_multi():
    """several independent batched transposes in one launch (kept for callers that need key-major copies)"""
    ops = _ops()
    a = rnd(3, 77, 64, seed=40).to(DEV)
    b = rnd(2, 130, 40, seed=41).to(DEV)
    at = torch.zeros(3, 64, 80, dtype=torch.float16, device=DEV)
    bt = torch.zeros(2, 40, 136, dtype=torch.float16, device=DEV)
    ops.transpose_multi([(a.view(-1, 64), at, 77, 64, 3, 64, 77 * 64, 80, 64 * 80),
                         (b.view(-1, 40), bt, 130, 40, 2, 40, 130 * 40, 136, 40 * 136)])
    torch.cuda.synchronize()
    assert torch.equal(at[:, :, :77], a.transpose(1, 2)) and torch.equal(bt[:, :, :130], b.transpose(1, 2))
    assert float(at[:, :, 77:].abs().max()) == 0 and float(bt[:, :, 130:].abs().max()) == 0
