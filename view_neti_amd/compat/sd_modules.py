"""Seam B of INTEGRATION.md for the two heaviest modules and the scheduler: drop-in objects for the *module-call protocol*
the reference's Coach uses (training/coach.py:165-205),

    latents = vae.encode(pixels).latent_dist.sample().detach() * vae.config.scaling_factor          (:165-169)
    noisy   = noise_scheduler.add_noise(latents, noise, timesteps)                                   (:182-183)
    pred    = unet(noisy, timesteps, _hs).sample             # _hs = {"this_idx", "CONTEXT_TENSOR_i", ..._BYPASS_i}   (:197-198)
    target  = noise | noise_scheduler.get_velocity(latents, noise, timesteps)                        (:201-205)
    F.mse_loss(pred.float(), target.float()).backward()                                              (:211-214)

served by the same launch schedules `TrainStepEngine` replays (engine/unet.py, engine/vae.py) and the same kernels
(`vneti_latent_sample`, `vneti_add_noise` round exactly like the fused `vneti_sample_add_noise`).  `HipUNet2DConditionModel`
is a `torch.autograd.Function` over `UNetEngine`: the UNet is frozen in the reference (coach.py:642-653), so its backward
returns gradients for the context tensors only (the XTI key contexts `CONTEXT_TENSOR_i` and value contexts
`CONTEXT_TENSOR_BYPASS_i`, models/xti_attention_processor.py:36-42) — none for the latents, which the reference detaches.

Static shapes: an adapter is built for one (batch, height, width); other shapes raise.  There is no fallback path: the
engines call the C ABI of libvneti_hip.so, whose absence raises at import of `view_neti_amd.lib`.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .. import lib, ops
from .. import sd_config as sc
from ..engine.step import alphas_cumprod
from ..engine.unet import UNetEngine
from ..engine.vae import VAEEncoderEngine


class _LatentDist:
    """`DiagonalGaussianDistribution` of diffusers 0.14 as far as the Coach uses it (`.sample()`; `.mode()`, `.mean`,
    `.logvar` for completeness)."""

    def __init__(self, owner: "HipAutoencoderKL"):
        self._o = owner
        # the moments are SNAPSHOT at encode() time (34 KB .. 0.5 MB): a later vae.encode() on the same static engine
        # must not change a distribution that was returned earlier
        self._mom = owner.engine.moments.clone()

    def _moments(self):
        o = self._o
        Lc = o.cfg.latent_channels
        m = self._mom.float().view(o.B, o.h, o.w, 2 * Lc).permute(0, 3, 1, 2)
        return m[:, :Lc], m[:, Lc:].clamp(-30.0, 20.0)

    @property
    def mean(self):
        return self._moments()[0].contiguous()

    @property
    def logvar(self):
        return self._moments()[1].contiguous()

    def mode(self):
        return self.mean

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        o = self._o
        Lc = o.cfg.latent_channels
        eps = torch.randn((o.B, Lc, o.h, o.w), generator=generator, device=o.dev, dtype=torch.float32)
        out = torch.empty_like(eps)
        ops.latent_sample(self._mom, eps, 1.0, out, o.B, Lc, o.h * o.w)
        o.last_eps = eps  # (tests: the same draw fed to TrainStepEngine.set_noise)
        return out


class HipAutoencoderKL:
    """`AutoencoderKL` (encoder side) behind `vae.encode(x).latent_dist.sample()` and `vae.config.scaling_factor`."""

    def __init__(self, cfg: sc.VAEConfig, weights: Dict[str, torch.Tensor], batch: int, height: int, width: int,
                 device: str = "cuda"):
        self.cfg, self.B, self.dev = cfg, batch, device
        self.engine = VAEEncoderEngine(cfg, weights, batch, height, width, device)
        self.h, self.w = self.engine.h_out, self.engine.w_out
        self.config = SimpleNamespace(scaling_factor=cfg.scaling_factor, latent_channels=cfg.latent_channels)
        self.dtype = lib.act_dtype()
        self.last_eps = None

    def to(self, *_, **__):  # coach.py:792-794 moves / casts the module; the engine already lives on the device in f16
        return self

    def requires_grad_(self, *_):
        return self

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        if tuple(x.shape) != tuple(self.engine.x_in.shape):
            raise ValueError(f"HipAutoencoderKL was built for pixels {tuple(self.engine.x_in.shape)}, got {tuple(x.shape)}")
        self.engine.x_in.copy_(x)
        self.engine.forward()
        return SimpleNamespace(latent_dist=_LatentDist(self))


class HipDDPMScheduler:
    """`DDPMScheduler(beta_schedule="scaled_linear")` as the Coach uses it: `.config`, `.add_noise`, `.get_velocity`."""

    def __init__(self, cfg: sc.DDPMConfig, device: str = "cuda"):
        self.config = SimpleNamespace(num_train_timesteps=cfg.num_train_timesteps, prediction_type=cfg.prediction_type,
                                      beta_start=cfg.beta_start, beta_end=cfg.beta_end, beta_schedule="scaled_linear")
        self.alphas_cumprod = alphas_cumprod(cfg).to(device)

    def _run(self, latents, noise, timesteps, want_noisy: bool, vpred: bool):
        if latents.dtype != torch.float32 or noise.dtype != torch.float32:
            latents, noise = latents.float(), noise.float()
        latents, noise = latents.contiguous(), noise.contiguous()
        B, Lc = latents.shape[0], latents.shape[1]
        hw = latents[0, 0].numel()
        out = torch.empty_like(latents)
        t = timesteps.to(device=latents.device, dtype=torch.int64).contiguous()
        ops.add_noise(latents, noise, t, self.alphas_cumprod, vpred, out if want_noisy else None,
                      None if want_noisy else out, B, Lc, hw)
        return out

    @torch.no_grad()
    def add_noise(self, original_samples, noise, timesteps):
        return self._run(original_samples, noise, timesteps, True, False)

    @torch.no_grad()
    def get_velocity(self, sample, noise, timesteps):
        return self._run(sample, noise, timesteps, False, True)


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, sample, timesteps, *contexts):
        eng = owner.engine
        nl = eng.nl
        eng.x_in.copy_(sample)
        eng.timesteps.copy_(timesteps)
        for i in range(nl):
            eng.ctx_k[i].copy_(contexts[i].reshape(eng.ctx_k[i].shape))
            eng.ctx_v[i].copy_(contexts[nl + i].reshape(eng.ctx_v[i].shape))
        eng.forward()
        # the engine holds ONE set of activations: a backward is only valid against the forward that filled them last
        owner._forward_count = getattr(owner, "_forward_count", 0) + 1
        ctx.forward_count = owner._forward_count
        ctx.owner = owner
        ctx.shapes = [c.shape for c in contexts]
        ctx.dtypes = [c.dtype for c in contexts]
        B, Co = eng.B, eng.cfg.out_channels
        return eng.pred.view(B, eng.H, eng.W, Co).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.owner.engine
        if not eng.need_backward:
            raise RuntimeError("HipUNet2DConditionModel was built with need_backward=False")
        if ctx.forward_count != ctx.owner._forward_count:
            raise RuntimeError(
                f"backward of UNet forward #{ctx.forward_count}, but the engine has since run forward "
                f"#{ctx.owner._forward_count}: the static engine keeps the activations of its LAST forward only (one "
                "forward per backward; run validation / prior-preservation forwards on a second engine)")
        nl = eng.nl
        B, Co = eng.B, eng.cfg.out_channels
        eng.dpred.copy_(grad_out.permute(0, 2, 3, 1).reshape(B * eng.H * eng.W, Co))
        eng.backward()
        grads = [eng.dctx_k[i].reshape(ctx.shapes[i]).to(ctx.dtypes[i]) for i in range(nl)]
        grads += [eng.dctx_v[i].reshape(ctx.shapes[nl + i]).to(ctx.dtypes[nl + i]) for i in range(nl)]
        return (None, None, None, *grads)


class HipUNet2DConditionModel:
    """`UNet2DConditionModel` with `XTIAttenProc` installed, behind `unet(sample, timesteps, _hs).sample`."""

    def __init__(self, cfg: sc.UNetConfig, weights: Dict[str, torch.Tensor], batch: int, height: int, width: int,
                 ctx_len: int = 77, device: str = "cuda", need_backward: bool = True):
        """height / width: of the LATENTS (pixels / 8)."""
        self.cfg, self.dev = cfg, device
        self.engine = UNetEngine(cfg, weights, batch, height, width, ctx_len, device, need_backward)
        self.config = SimpleNamespace(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                      cross_attention_dim=cfg.cross_attention_dim)
        self.dtype = lib.act_dtype()

    def to(self, *_, **__):
        return self

    def requires_grad_(self, *_):
        return self

    def set_attn_processor(self, _proc):
        """coach.py:679-680 installs XTIAttenProc on every attention module; the engine's attention layers ARE that
        processor (K from CONTEXT_TENSOR_i, V from CONTEXT_TENSOR_BYPASS_i, one context pair per layer in call order)."""

    def __call__(self, sample: torch.Tensor, timesteps: torch.Tensor, encoder_hidden_states: Dict, **_):
        eng = self.engine
        if not isinstance(encoder_hidden_states, dict):
            raise TypeError("HipUNet2DConditionModel takes the XTI context dict (prompt_manager.py:79-99 / coach.py:287-305)")
        if tuple(sample.shape) != tuple(eng.x_in.shape):
            raise ValueError(f"HipUNet2DConditionModel was built for latents {tuple(eng.x_in.shape)}, got {tuple(sample.shape)}")
        hs = encoder_hidden_states
        ks = [hs[f"CONTEXT_TENSOR_{i}"] for i in range(eng.nl)]
        # xti_attention_processor.py:39-42: without a bypass tensor the value context is the key context
        vs = [hs.get(f"CONTEXT_TENSOR_BYPASS_{i}", ks[i]) for i in range(eng.nl)]
        out = _UNetFn.apply(self, sample, timesteps, *ks, *vs)
        hs["this_idx"] = 0  # the processor's layer counter wraps after the last attention layer (:21-23,53-55)
        return SimpleNamespace(sample=out)

    forward = __call__
