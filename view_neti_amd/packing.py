"""Weight packing: torch/diffusers-layout parameters -> the layouts the HIP kernels consume.

conv2d weight [Co][Ci][3][3]  -> forward  B[Co][tap*Ci + ci]          (K = 9*Ci contiguous)
                              -> dgrad    B[Ci][tap*Co + co]          (transposed gather, unflipped taps)
linear weight [out][in]       -> forward  as is ([N][K]); dgrad = its transpose [in][out]
1x1 conv weight [Co][Ci][1][1] is a linear weight.
GEGLU projection [2*F][in] (rows 0..F-1 = h, F..2F-1 = gate) -> rows interleaved in groups of four,
                              [h0..h3 g0..g3 h4..h7 g4..g7 ...], so that one 16-byte chunk of the GEMM's output tile
                              holds matching halves and the epilogue can apply h * gelu(g) (vneti_gemm_desc.geglu).
"""
from __future__ import annotations

import os

import torch

# K order of the implicit-GEMM convolutions (vneti_gemm_desc.conv_korder).  0: (tap, channel) — a k-step walks the
# channels of one tap, so the nine re-reads of an input pixel are a whole channel sweep apart (>= 160 KB per block, past
# the XCD's L2 at 32 blocks per XCD).  1: (64-channel chunk, tap, channel-in-chunk) — the nine taps of a chunk are
# consecutive k-steps and re-read the same few KB straight from L1/L2.  Only for channel counts that are multiples of
# 64 (the others go through vneti_im2col3x3_small and keep the tap-major order its output has).
# Measured (r02, bench A/B in one box): chunk-major is SLOWER on the step, 32.8 vs 31.8 ms — every k-step recomputes the
# gather addresses and the autotuner falls back to smaller tiles; the L2 re-reads were not the limiter.  Kept switchable.
KORDER_CM = os.environ.get("VNETI_CONV_KORDER", "0") == "1"


def _chunk_major(m: torch.Tensor, n_out: int, c: int) -> torch.Tensor:
    """[n_out][9][c] -> [n_out][c/64][9][64] flattened"""
    return m.reshape(n_out, 9, c // 64, 64).permute(0, 2, 1, 3).reshape(n_out, 9 * c).contiguous()


def conv3x3_fwd(w: torch.Tensor, cm=None) -> torch.Tensor:
    """cm: chunk-major K order (None = the module default KORDER_CM); needs Ci % 64 == 0"""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    m = w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()
    cm = KORDER_CM if cm is None else cm
    return _chunk_major(m, co, ci) if cm and ci % 64 == 0 else m


def conv3x3_dgrad(w: torch.Tensor, cm=None) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    m = w.permute(1, 2, 3, 0).reshape(ci, 9 * co).contiguous()
    cm = KORDER_CM if cm is None else cm
    return _chunk_major(m, ci, co) if cm and co % 64 == 0 else m


def pad_rows(w: torch.Tensor, mult: int) -> torch.Tensor:
    """zero-pad the K (last) dim to a multiple of `mult`."""
    k = w.shape[-1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros(*w.shape[:-1], kp, dtype=w.dtype, device=w.device)
    out[..., :k] = w
    return out


def geglu_interleave_index(n2: int, device=None) -> torch.Tensor:
    """row permutation of a GEGLU projection with 2*F = n2 outputs: position 8q+e takes h[4q+e] (e < 4) or
    gate[4q+e-4] (e >= 4)"""
    F = n2 // 2
    assert n2 % 8 == 0
    r = torch.arange(n2, device=device)
    q, e = r // 8, r % 8
    return torch.where(e < 4, 4 * q + e, F + 4 * q + e - 4)


def geglu_interleave(w: torch.Tensor) -> torch.Tensor:
    """weight [2F][in] or bias [2F] -> the interleaved row order"""
    return w[geglu_interleave_index(w.shape[0], w.device)].contiguous()


def conv_in_direct(w: torch.Tensor) -> torch.Tensor:
    """[Co, C <= 3, 3, 3] -> [Co][32] for vneti_conv3x3_in: k = tap * C + c (zeros from 9 * C up); inside every 128-channel
    block, row j * 16 + 4 * fq + e holds output channel (j // 2) * 32 + fq * 8 + (j % 2) * 4 + e, so that after the swapped
    16x16x32 MFMA a lane owns 8 consecutive channels of each 32-channel chunk (csrc/conv_in.hip)."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3 and 9 * ci <= 32 and co % 128 == 0
    m = torch.zeros(co, 32, dtype=w.dtype)
    m[:, : 9 * ci] = w.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    j = torch.arange(8).view(8, 1, 1)
    fq = torch.arange(4).view(1, 4, 1)
    e = torch.arange(4).view(1, 1, 4)
    src = ((j // 2) * 32 + fq * 8 + (j % 2) * 4 + e).reshape(128)      # packed row j*16 + 4*fq + e  <-  channel src
    idx = (torch.arange(co // 128).view(-1, 1) * 128 + src.view(1, 128)).reshape(-1)
    return m[idx].contiguous()
