"""Weight packing: torch/diffusers-layout parameters -> the layouts the HIP kernels consume.

conv2d weight [Co][Ci][3][3]  -> forward  B[Co][tap*Ci + ci]          (K = 9*Ci contiguous)
                              -> dgrad    B[Ci][tap*Co + co]          (transposed gather, unflipped taps)
linear weight [out][in]       -> forward  as is ([N][K]); dgrad = its transpose [in][out]
1x1 conv weight [Co][Ci][1][1] is a linear weight.
"""
from __future__ import annotations

import torch


def conv3x3_fwd(w: torch.Tensor) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    return w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()


def conv3x3_dgrad(w: torch.Tensor) -> torch.Tensor:
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    return w.permute(1, 2, 3, 0).reshape(ci, 9 * co).contiguous()


def pad_rows(w: torch.Tensor, mult: int) -> torch.Tensor:
    """zero-pad the K (last) dim to a multiple of `mult`."""
    k = w.shape[-1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros(*w.shape[:-1], kp, dtype=w.dtype, device=w.device)
    out[..., :k] = w
    return out
