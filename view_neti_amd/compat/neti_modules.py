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


class NeTIPositionalEncoding(nn.Module):
    """models/positional_encoding.py:10-51 — the LEGACY (t, l) encoder of `arch_view_net <= 14` (the dataclass default):
    num_w = 1024 random frequencies drawn from the GLOBAL generator (no seed: the pickled instance inside a checkpoint is
    the only record of them), v = normalised cat[sin(w x), cos(w x)] of the RAW timestep / layer index."""

    def __init__(self, sigma_t: float, sigma_l: float, num_w: int = 1024):
        super().__init__()
        self.sigma_t, self.sigma_l, self.num_w = sigma_t, sigma_l, num_w
        w = torch.randn((num_w, 2))
        w[:, 0] *= sigma_t
        w[:, 1] *= sigma_l
        self.w = w  # plain tensor attribute, as on a CUDA run of the reference (App. C Q2)

    def encode(self, t, l):
        if isinstance(t, (int, float)) or getattr(t, "ndim", 1) == 0:
            x = torch.tensor([float(t), float(l)])
            v = torch.cat([torch.sin(self.w @ x), torch.cos(self.w @ x)])
            return v / v.norm()
        x = torch.stack([t.float(), l.float()], dim=1).t().to(self.w.device)
        v = torch.cat([torch.sin(self.w @ x), torch.cos(self.w @ x)])
        return (v / v.norm(dim=0)).t()

    def init_layer(self, num_time_anchors: int, num_layers: int) -> torch.Tensor:
        return torch.stack([self.encode(t_anchor, l_anchor).float()
                            for t_anchor in range(0, 1000, 1000 // num_time_anchors) for l_anchor in range(num_layers)])


NeTIPositionalEncoding.__module__ = "models.positional_encoding"


class NeTIMapper(nn.Module):
    """models/neti_mapper.py:19-611: the paper's arch_view_net = 15 mapper — Fourier(t, l[, 12 camera params]) ->
    Linear-LN-LeakyReLU x2 -> Linear -> [word | bypass] — and the legacy object mapper of arch_view_net <= 14 (the
    dataclass default): NeTIPositionalEncoding -> anchor-initialised input_layer -> the same MLP with h hidden units."""

    def __init__(self, embedding_type: str = "object", output_dim: int = 768, arch_mlp_hidden_dims: int = 64,
                 norm_scale: Optional[float] = None, pe_sigmas=None, output_bypass: bool = True,
                 bypass_unconstrained: bool = False, output_bypass_alpha: float = 0.2,
                 placeholder_object_token: Optional[str] = None, cam_mins: Optional[torch.Tensor] = None,
                 cam_maxs: Optional[torch.Tensor] = None, num_unet_layers: int = 16,
                 use_nested_dropout: bool = False, nested_dropout_prob: float = 0.5, arch_view_net: int = 15,
                 num_pe_time_anchors: int = 10):
        super().__init__()
        assert embedding_type in ("object", "view")
        self.embedding_type = embedding_type
        self.arch_view_net = arch_view_net
        self.legacy = arch_view_net <= 14
        if self.legacy and embedding_type == "view":
            raise NotImplementedError("the legacy view mapper needs `encode_phi`, which the reference never defines "
                                      "(neti_mapper.py:347-348): arch_view_net <= 14 works for object mappers only")
        self.output_bypass = output_bypass
        self.bypass_unconstrained = bypass_unconstrained
        self.output_bypass_alpha = output_bypass_alpha
        self.norm_scale = float(norm_scale) if norm_scale is not None else None
        self.placeholder_object_token = placeholder_object_token
        self.num_unet_layers = num_unet_layers
        self.use_nested_dropout = use_nested_dropout
        self.nested_dropout_prob = nested_dropout_prob
        st, sl = (pe_sigmas.sigma_t, pe_sigmas.sigma_l) if pe_sigmas is not None else (0.03, 2.0)
        if self.legacy:
            # neti_mapper.py:90-163: NeTIPositionalEncoding -> input_layer initialised from the anchors -> MLP(h)
            h = arch_mlp_hidden_dims
            n_in = num_pe_time_anchors * num_unet_layers
            self.encoder = NeTIPositionalEncoding(st, sl)
            self.input_layer = nn.Linear(self.encoder.num_w * 2, n_in)
            self.input_layer.weight.data = self.encoder.init_layer(num_pe_time_anchors, num_unet_layers)
            out = output_dim * 2 if output_bypass else output_dim
            self.net = nn.Sequential(nn.Linear(n_in, h), nn.LayerNorm(h), nn.LeakyReLU(), nn.Linear(h, h),
                                     nn.LayerNorm(h), nn.LeakyReLU())
            self.output_layer = nn.Sequential(nn.Linear(h, out))
            self.hidden, self.enc_dim = h, n_in
            return
        self.enc_dim = 64
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
        if self.legacy:
            y = self.output_layer(self.net(self.input_layer(self.encoder.encode(timestep, unet_layer))))
            d = y.shape[1] // 2
            word, byp = (y[:, :d], y[:, d:]) if self.output_bypass else (y, None)
            if self.norm_scale is not None:
                word = F.normalize(word, dim=-1) * self.norm_scale
            return word, byp
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

    def engine_encoder_kwargs(self):
        """what TrainStepEngine / InferenceEngine need to know about this (object) mapper's encoder"""
        kw = dict(output_bypass_object=self.output_bypass)
        if self.legacy:
            kw.update(legacy_pe_object=self.encoder.w, enc_dim_object=self.enc_dim)
        return kw

    @property
    def pe_dim(self) -> int:
        return 2 * self.encoder.num_w if self.legacy else 0

    def mapper_state(self):
        """state_dict without the encoder (what the reference saves/loads with strict=True)."""
        return {k: v for k, v in self.state_dict().items() if not k.startswith("encoder")}


NeTIMapper.__module__ = "models.neti_mapper"
