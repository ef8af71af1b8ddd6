"""torch.nn.Module views of the NeTI mapper for checkpoint interchange with the reference.

The HIP engine trains a flat fp32 bucket; on disk the reference expects (checkpoint_handler.py:57-97)
  {"cfg": <encoded RunConfig>, "mappers": {token_id: {"state_dict": ..., "encoder": <pickled nn.Module>,
                                                     "placeholder_object_token": str}}}
with state_dict keys net.0/1/3/4.{weight,bias}, output_layer.0.{weight,bias} and NO `encoder.w`
(SURVEY App. C Q2).  The pickled encoder resolves its class by module path, so the class below
advertises `models.positional_encoding` / `models.neti_mapper` as its module and the repo ships thin alias
modules under those names.  These modules are device-agnostic (no hard .cuda()).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from ..mapper import fourier_frequencies


class FourierPositionalEncodingNDims(nn.Module):
    """models/positional_encoding.py:146-195.  `w` is a plain tensor attribute (neither Parameter nor
    buffer): it never appears in state_dict and is regenerated from `seed` (torch.manual_seed side
    effect included, to keep the reference's RNG stream)."""

    def __init__(self, sigmas: List[float], dim: int = 128, normalize: bool = False, seed: int = 0):
        super().__init__()
        self.sigmas = sigmas
        self.dim = dim
        self.normalize = normalize
        self.w = fourier_frequencies(sigmas, dim, seed)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.ndim == 1:
            x = x.unsqueeze(1)
        proj = self.w.detach().to(x.device) @ x.t()
        v = torch.cat([torch.sin(proj), torch.cos(proj)])
        if self.normalize:
            v = v / v.norm(dim=0)
        return v.t()


FourierPositionalEncodingNDims.__module__ = "models.positional_encoding"


class NeTIMapper(nn.Module):
    """arch_view_net = 15 mapper (models/neti_mapper.py:19-611 restricted to the paper's architecture):
    Fourier(t, l[, 12 camera params]) -> Linear-LN-LeakyReLU x2 -> Linear -> [word | bypass]."""

    def __init__(self, embedding_type: str = "object", output_dim: int = 768, arch_mlp_hidden_dims: int = 64,
                 norm_scale: Optional[float] = None, pe_sigmas=None, output_bypass: bool = True,
                 bypass_unconstrained: bool = False, output_bypass_alpha: float = 0.2,
                 placeholder_object_token: Optional[str] = None, cam_mins: Optional[torch.Tensor] = None,
                 cam_maxs: Optional[torch.Tensor] = None, num_unet_layers: int = 16,
                 use_nested_dropout: bool = False, nested_dropout_prob: float = 0.5):
        super().__init__()
        assert embedding_type in ("object", "view")
        self.embedding_type = embedding_type
        self.arch_view_net = 15
        self.output_bypass = output_bypass
        self.bypass_unconstrained = bypass_unconstrained
        self.output_bypass_alpha = output_bypass_alpha
        self.norm_scale = float(norm_scale) if norm_scale is not None else None
        self.placeholder_object_token = placeholder_object_token
        self.num_unet_layers = num_unet_layers
        self.use_nested_dropout = use_nested_dropout
        self.nested_dropout_prob = nested_dropout_prob
        st, sl = (pe_sigmas.sigma_t, pe_sigmas.sigma_l) if pe_sigmas is not None else (0.03, 2.0)
        sigmas = [st, sl]
        if embedding_type == "view":
            sigmas += [pe_sigmas.sigma_dtu12 if pe_sigmas is not None else 0.5] * 12
            self.cam_mins, self.cam_maxs = cam_mins, cam_maxs
        # order matters: the encoder re-seeds the global RNG, then the Linears draw from it (Q1)
        self.encoder = FourierPositionalEncodingNDims(dim=64, sigmas=sigmas, seed=0)
        h = arch_mlp_hidden_dims if embedding_type == "object" else 64  # neti_mapper.py:148 vs :603
        out = output_dim * 2 if output_bypass else output_dim
        self.net = nn.Sequential(nn.Linear(64, h), nn.LayerNorm(h), nn.LeakyReLU(), nn.Linear(h, h), nn.LayerNorm(h),
                                 nn.LeakyReLU())
        self.output_layer = nn.Sequential(nn.Linear(h, out))
        self.hidden = h

    def forward(self, timestep, unet_layer, view_params: Optional[torch.Tensor] = None):
        data = torch.stack((timestep.float() / 1000 * 2 - 1, unet_layer.float() / self.num_unet_layers * 2 - 1), dim=1)
        if self.embedding_type == "view":
            data = torch.cat((data, view_params.to(data)), dim=1)
        y = self.output_layer(self.net(self.encoder(data)))
        if self.output_bypass:
            d = y.shape[1] // 2
            word, byp = y[:, :d], y[:, d:]
        else:
            word, byp = y, None
        if self.norm_scale is not None:
            word = F.normalize(word, dim=-1) * self.norm_scale
        return word, byp

    def mapper_state(self):
        """state_dict without the encoder (what the reference saves/loads with strict=True)."""
        return {k: v for k, v in self.state_dict().items() if not k.startswith("encoder")}


NeTIMapper.__module__ = "models.neti_mapper"
