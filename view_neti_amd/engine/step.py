"""One textual-inversion optimisation step on one GPU as a single replayable launch schedule.

Mirrors the loop body of `Coach.train` (training/coach.py:154-231):

    latents = vae.encode(px).latent_dist.sample() * scaling_factor        -> VAEEncoderEngine + sample_add_noise
    noise, timesteps, noisy = randn_like, randint, scheduler.add_noise    -> device RNG + sample_add_noise
    _hs = get_text_conditioning(...)                                      -> TextEngine.forward
    model_pred = unet(noisy, timesteps, _hs).sample                       -> UNetEngine.forward
    loss = mse_loss(model_pred.float(), target.float())                   -> mse_loss_grad
    accelerator.backward(loss)  (GradScaler-scaled under fp16)            -> UNetEngine.backward + TextEngine.backward
    optimizer.step(); zero_grad()                                         -> adamw_flat (flat bucket)

Everything that varies per step lives in device memory (RNG counter, optimizer step, loss scale,
learning rate), so the step is captured once into a hipGraph and replayed.  Deliberate
deviations from the reference, all documented in DESIGN.md: the placeholder-embedding "restore"
(coach.py:222-229) is dropped (the table is never in the optimizer); noise/timesteps come from a
counter-hash device RNG instead of torch's Philox stream; data-parallel training all-reduces one
flat mapper-gradient bucket instead of wrapping the text encoder in DDP.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .. import ops
from .. import sd_config as sc
from .text import MapperState, TextEngine
from .unet import UNetEngine
from .vae import VAEEncoderEngine


def alphas_cumprod(cfg: sc.DDPMConfig) -> torch.Tensor:
    """DDPMScheduler(beta_schedule='scaled_linear'): betas = linspace(sqrt(b0), sqrt(b1), T)^2."""
    betas = torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, cfg.num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class TrainStepEngine:
    def __init__(self, cfg: sc.SDConfig, unet_w: Dict, vae_w: Dict, clip_w: Dict, batch: int, height: int,
                 width: int, mapper_object: Dict[str, torch.Tensor], w_enc_object: torch.Tensor,
                 norm_scale_object: Optional[float], alpha_object: float = 0.2,
                 mapper_view: Optional[Dict[str, torch.Tensor]] = None, w_enc_view: Optional[torch.Tensor] = None,
                 norm_scale_view: Optional[float] = None, alpha_view: float = 0.2, train_view: bool = True,
                 n_view_params: int = 12, lr: float = 1e-3, betas=(0.9, 0.999), adam_eps: float = 1e-8,
                 weight_decay: float = 1e-2, loss_scale: float = 65536.0, growth_interval: int = 2000,
                 seed: int = 0, world_size: int = 1, device_rng: bool = True, device: str = "cuda",
                 need_backward: bool = True, grad_accum: int = 1):
        from .text import flatten_mapper_state
        self.cfg = cfg
        self.B, self.H, self.W = batch, height, width
        self.dev = device
        self.world_size = world_size
        self.device_rng = device_rng
        self.growth_interval = growth_interval
        nlev = len(cfg.vae.block_out_channels)
        self.h, self.w = height >> (nlev - 1), width >> (nlev - 1)
        Lc = cfg.vae.latent_channels
        D = cfg.clip.hidden_size
        # ---- trainable state: one flat f32 bucket (object mapper [+ view mapper]) ----
        flat_o = flatten_mapper_state(mapper_object)
        flat_v = flatten_mapper_state(mapper_view) if (mapper_view is not None and train_view) else None
        n = flat_o.numel() + (flat_v.numel() if flat_v is not None else 0)
        self.params = torch.zeros(n, dtype=torch.float32, device=device)
        self.params[: flat_o.numel()].copy_(flat_o)
        if flat_v is not None:
            self.params[flat_o.numel():].copy_(flat_v)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.n_obj = flat_o.numel()
        mo = MapperState(self.params[: self.n_obj], w_enc_object.to(device).float().contiguous(), norm_scale_object,
                         alpha_object)
        mv, gv = None, None
        if mapper_view is not None:
            if flat_v is not None:
                pv, gv = self.params[self.n_obj:], self.grads[self.n_obj:]
            else:  # frozen pretrained view mapper (learnable_mode 4/5)
                pv = flatten_mapper_state(mapper_view).to(device)
            mv = MapperState(pv, w_enc_view.to(device).float().contiguous(), norm_scale_view, alpha_view)
        # ---- device-resident scalars ----
        self.grad_accum = grad_accum
        # accelerate scales each micro-loss by 1/accum and DDP averages over ranks: fold both into AdamW
        self.hyper = torch.tensor([lr, betas[0], betas[1], adam_eps, weight_decay, float(world_size * grad_accum)],
                                  dtype=torch.float32, device=device)
        self.scaler = torch.tensor([loss_scale, 0.0, 0.0], dtype=torch.float32, device=device)
        self.opt_step = torch.zeros(1, dtype=torch.int32, device=device)
        self.rng_state = torch.tensor([seed & 0x7FFFFFFF, 0], dtype=torch.int32, device=device)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
        self.ac = alphas_cumprod(cfg.ddpm).to(device)
        # ---- engines ----
        self.unet = UNetEngine(cfg.unet, unet_w, batch, self.h, self.w, cfg.clip.max_positions, device, need_backward)
        self.vae = VAEEncoderEngine(cfg.vae, vae_w, batch, height, width, device)
        self.text = TextEngine(cfg.clip, clip_w, cfg.unet.n_cross_layers, batch, self.unet.timesteps, self.unet.ctx_k,
                               self.unet.ctx_v, self.unet.dctx_k, self.unet.dctx_v, mo, self.grads[: self.n_obj], mv,
                               gv, n_view_params, train_view, device, need_backward)
        self.timesteps = self.unet.timesteps
        self.pixel_values = self.vae.x_in
        shape = (batch, Lc, self.h, self.w)
        self.eps = torch.zeros(shape, dtype=torch.float32, device=device)
        self.noise = torch.zeros(shape, dtype=torch.float32, device=device)
        self.latents = torch.zeros(shape, dtype=torch.float32, device=device)
        self.target = torch.zeros(shape, dtype=torch.float32, device=device)
        self.need_backward = need_backward
        self.graph_a = self.graph_b = self.graph_acc = None
        self.micro = 0
        self.n_loss = batch * Lc * self.h * self.w

    # ------------------------------------------------------------------ inputs
    def set_batch(self, pixel_values, input_ids, placeholder_object, placeholder_view=None, view_params=None):
        self.pixel_values.copy_(pixel_values, non_blocking=True)
        self.text.set_batch(input_ids, placeholder_object, placeholder_view, view_params)

    def set_noise(self, eps, noise, timesteps):
        """host-supplied randomness (parity tests: identical values for the oracle and the GPU)."""
        self.eps.copy_(eps)
        self.noise.copy_(noise)
        self.timesteps.copy_(timesteps)

    def set_lr(self, lr: float):
        self.hyper[0] = lr

    # ------------------------------------------------------------------ the step
    def forward_backward(self, accumulate: bool = False):
        B, Lc, hw = self.B, self.cfg.vae.latent_channels, self.h * self.w
        self.text.accumulate_grads = accumulate
        if self.device_rng:
            ops.rng_advance(self.rng_state)
            ops.rng_fill_randint(self.timesteps, self.cfg.ddpm.num_train_timesteps, self.rng_state, 0)
            ops.rng_fill_normal(self.eps, self.rng_state, 1)
            ops.rng_fill_normal(self.noise, self.rng_state, 2)
        self.vae.forward()
        ops.sample_add_noise(self.vae.moments, self.eps, self.noise, self.timesteps, self.ac,
                             self.cfg.vae.scaling_factor, self.cfg.ddpm.prediction_type == "v_prediction", self.latents,
                             self.unet.x_in, self.target, B, Lc, hw)
        self.text.forward()
        self.unet.forward()
        self.loss_sum.zero_()
        ops.mse_loss_grad(self.unet.pred, self.target, self.unet.dpred, self.loss_sum, self.scaler, B, Lc, hw)
        if self.need_backward:
            self.unet.backward()
            self.text.backward()

    def optimizer_step(self):
        ops.adamw_flat(self.params, self.grads, self.exp_avg, self.exp_avg_sq, self.hyper, self.scaler, self.opt_step,
                       self.growth_interval)

    def all_reduce(self):
        if self.world_size > 1:
            from ..parallel import all_reduce_sum_
            all_reduce_sum_(self.grads)

    def step_eager(self):
        """one micro-step; the optimizer runs after every `grad_accum`-th micro-step."""
        self.forward_backward(accumulate=self.micro > 0)
        self.micro += 1
        if self.micro == self.grad_accum:
            self.micro = 0
            self.all_reduce()
            self.optimizer_step()
            return True
        return False

    # ------------------------------------------------------------------ hipGraph capture
    def capture(self):
        """Capture the step into hipGraphs: [forward+backward] and [optimizer], with the RCCL
        all-reduce of the flat gradient bucket between them (single graph when world_size == 1)."""
        # the warm-up launches below are real steps: snapshot the trainable / RNG state and put it back
        state = [t.clone() for t in (self.params, self.exp_avg, self.exp_avg_sq, self.opt_step, self.scaler,
                                     self.rng_state)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.grad_accum):
                self.step_eager()  # warm-up on the side stream (also primes RCCL)
            torch.cuda.synchronize()
            fused_opt = self.world_size == 1 and self.grad_accum == 1
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, stream=s):
                self.forward_backward(accumulate=False)
                if fused_opt:
                    self.optimizer_step()
            if self.grad_accum > 1:
                self.graph_acc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_acc, stream=s):
                    self.forward_backward(accumulate=True)
            if not fused_opt:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, stream=s):
                    self.optimizer_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for dst, src in zip((self.params, self.exp_avg, self.exp_avg_sq, self.opt_step, self.scaler, self.rng_state),
                            state):
            dst.copy_(src)
        self.micro = 0

    def step(self) -> bool:
        """one micro-step (graph replay when captured); returns True when the optimizer stepped."""
        if self.graph_a is None:
            return self.step_eager()
        (self.graph_a if self.micro == 0 else self.graph_acc).replay()
        self.micro += 1
        if self.micro < self.grad_accum:
            return False
        self.micro = 0
        if self.graph_b is not None:
            self.all_reduce()
            self.graph_b.replay()
        return True

    def loss(self) -> float:
        """mean squared error of the last step (forces a device sync — call sparingly)."""
        return float(self.loss_sum.item()) / self.n_loss

    # ------------------------------------------------------------------ introspection
    def launches(self) -> List:
        out = list(self.vae.fwd) + list(self.text.fwd) + list(self.unet.fwd)
        if self.need_backward:
            out += list(self.unet.bwd) + list(self.text.bwd)
        return out

    def memory_bytes(self) -> int:
        return self.unet.bytes + self.vae.bytes + self.text.bytes
