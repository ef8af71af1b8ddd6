"""Host-side weight packing (view_neti_amd/packing.py) against plain torch on the CPU: the layouts the HIP kernels assume.

  * conv3x3_fwd / conv3x3_dgrad: the implicit-GEMM K order (tap, channel) and the flipped, transposed dgrad weights
    (diffusers Conv2d layers driven from training/coach.py:165-169, 197-198)
  * geglu_interleave: the [h0..3 g0..3 h4..7 ...] row order of ff.net.0.proj (vneti_gemm_desc.geglu)
  * conv_in_direct: the [Co][32] operand of vneti_conv3x3_in with its per-128-channel row permutation
"""
import torch
import torch.nn.functional as F

from view_neti_amd import packing


def _rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_conv3x3_fwd_is_im2col_times_packed():
    x, w = _rnd(2, 8, 6, 7, seed=1), _rnd(5, 8, 3, 3, seed=2)
    ref = F.conv2d(x, w, padding=1)
    cols = F.unfold(x, 3, padding=1)                                  # [B, C*9, HW] with k = c*9 + tap
    cols = cols.view(2, 8, 9, -1).permute(0, 3, 2, 1).reshape(2, 42, 72)  # -> k = tap*C + c
    out = (cols @ packing.conv3x3_fwd(w, cm=False).t()).permute(0, 2, 1).reshape(2, 5, 6, 7)
    assert torch.allclose(out, ref, atol=1e-4)


def test_conv3x3_dgrad_is_the_transposed_convolution():
    w, dy = _rnd(5, 8, 3, 3, seed=3), _rnd(2, 5, 6, 7, seed=4)
    x = _rnd(2, 8, 6, 7, seed=5).requires_grad_(True)
    F.conv2d(x, w, padding=1).backward(dy)
    wd = packing.conv3x3_dgrad(w, cm=False)                           # [Ci][9*Co]: a forward conv of dy with flipped taps
    cols = F.unfold(dy, 3, padding=1).view(2, 5, 9, -1).permute(0, 3, 2, 1).reshape(2, 42, 45)
    # the kernel gathers dy at (y + pad - dy_tap): as a forward conv that is the spatially flipped kernel
    cols_flip = cols.view(2, 42, 9, 5).flip(2).reshape(2, 42, 45)
    dx = (cols_flip @ wd.t()).permute(0, 2, 1).reshape(2, 8, 6, 7)
    assert torch.allclose(dx, x.grad, atol=1e-4)


def test_geglu_interleave_round_trip():
    C = 16
    w = _rnd(2 * C, 6, seed=6)
    il = packing.geglu_interleave(w)
    idx = packing.geglu_interleave_index(2 * C)
    assert torch.equal(il, w[idx])
    h, g = w[:C], w[C:]
    for blk in range(C // 4):
        assert torch.equal(il[8 * blk:8 * blk + 4], h[4 * blk:4 * blk + 4])
        assert torch.equal(il[8 * blk + 4:8 * blk + 8], g[4 * blk:4 * blk + 4])


def test_conv_in_direct_layout():
    """packed row j*16 + 4*fq + e of a 128-channel block holds channel (j//2)*32 + fq*8 + (j%2)*4 + e with k = tap*C + c;
    emulating the kernel's MFMA ownership (lane (pixel, fq) gets rows 4*fq..4*fq+3 of row block j) must give conv2d"""
    Co, C, H, W = 256, 3, 5, 6
    w, x = _rnd(Co, C, 3, 3, seed=7), _rnd(2, C, H, W, seed=8)
    pk = packing.conv_in_direct(w)
    assert pk.shape == (Co, 32) and torch.count_nonzero(pk[:, 9 * C:]) == 0
    cols = F.unfold(x, 3, padding=1).view(2, C, 9, -1).permute(0, 3, 2, 1).reshape(2, H * W, 9 * C)
    cols = F.pad(cols, (0, 32 - 9 * C))
    acc = cols @ pk.t()                                               # [B, HW, packed row]
    out = torch.empty(2, H * W, Co)
    for nb in range(Co // 128):
        for j in range(8):
            for fq in range(4):
                for e in range(4):
                    out[:, :, nb * 128 + (j // 2) * 32 + fq * 8 + (j % 2) * 4 + e] = acc[:, :, nb * 128 + j * 16 + 4 * fq + e]
    ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1).reshape(2, H * W, Co)
    assert torch.allclose(out, ref, atol=1e-4)
