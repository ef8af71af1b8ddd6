"""Host-side pieces of the NeTI mapper that are not kernels: Fourier frequency generation with
the reference's RNG quirk, parameter initialisation, and (un)flattening to the kernel bucket.

Reference: models/positional_encoding.py:146-172 (FourierPositionalEncodingNDims.__init__ calls
torch.manual_seed(seed) — a GLOBAL RNG side effect that makes `w` reproducible and is why `w` is
not in the state_dict) and models/neti_mapper.py:138-153,470-540,580-611.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch


def fourier_frequencies(sigmas: List[float], dim: int = 64, seed: int = 0, preserve_rng: bool = False) -> torch.Tensor:
    """w[dim/2, nfeats] ~ N(0,1) * sigma_i, drawn right after torch.manual_seed(seed).
    With preserve_rng=False this reproduces the reference's global side effect exactly."""
    state = torch.random.get_rng_state() if preserve_rng else None
    torch.manual_seed(seed)
    w = torch.randn((dim // 2, len(sigmas)))
    for i, s in enumerate(sigmas):
        w[:, i] *= s
    if state is not None:
        torch.random.set_rng_state(state)
    return w


def init_mapper_state(input_dim: int, hidden: int, output_dim: int, output_bypass: bool = True) -> Dict[str, torch.Tensor]:
    """nn.Linear / nn.LayerNorm default initialisation in module-construction order
    (net.0, net.1, net.3, net.4, output_layer.0), drawing from the current global torch RNG like
    the reference's nn.Sequential construction does (neti_mapper.py:148-153)."""
    od = output_dim * 2 if output_bypass else output_dim

    def linear(o, i):
        w = torch.empty(o, i)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1 / math.sqrt(i)
        b = torch.empty(o).uniform_(-bound, bound)
        return w, b

    sd = {}
    sd["net.0.weight"], sd["net.0.bias"] = linear(hidden, input_dim)
    sd["net.1.weight"], sd["net.1.bias"] = torch.ones(hidden), torch.zeros(hidden)
    sd["net.3.weight"], sd["net.3.bias"] = linear(hidden, hidden)
    sd["net.4.weight"], sd["net.4.bias"] = torch.ones(hidden), torch.zeros(hidden)
    sd["output_layer.0.weight"], sd["output_layer.0.bias"] = linear(od, hidden)
    return sd
