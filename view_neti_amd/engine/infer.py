"""Inference with learned NeTI mappers on the HIP engines: the loop of the reference's
`sd_pipeline_call` (sd_pipeline_call.py:8-133) with `PromptManager.embed_prompt`'s per-timestep, per-layer
text conditioning (prompt_manager.py:43-101) computed inside the loop.

Per denoising step i (timestep t_i):
    contexts   = 16 x [NeTI mapper(t_i, layer) -> CLIP -> bypass -> final LN]      TextEngine.forward (one batched pass)
    [eps_u; eps_c] = UNet([x; x], t_i, [uncond ctx; NeTI ctx])                      one CFG-batched UNetEngine.forward
    x <- sampler(x, eps_u + g (eps_c - eps_u))                                      vneti_cfg_sampler_step
then  image = (decode(x / scaling)/2 + 0.5).clamp(0,1)                              VAEDecoderEngine.forward

What differs from the reference on purpose: the two UNet calls of a step run as one batch of 2B (the reference
runs them back to back, :78-94); the T x 16 text-encoder passes are not materialised up front (the reference
holds T dicts of 32 tensors) but produced per step, 16 layers at a time; the unconditional embedding is computed
once.  Samplers: DPM-Solver++(2M) (the scheduler validate.py:568 / inference_dtu.py:304 install) and DDIM
(eta 0), both as x <- cx x + c0 x0 + c1 x0_prev on the data prediction.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .. import lib, ops
from .. import sd_config as sc
from .step import alphas_cumprod
from .text import MapperState, TextEngine, flatten_mapper_state
from .unet import UNetEngine
from .vae import VAEDecoderEngine


def inference_timesteps(kind: str, num_steps: int, num_train: int = 1000) -> List[int]:
    """DPMSolverMultistepScheduler.set_timesteps: linspace(0, T-1, N+1).round()[::-1][:-1];
    DDIMScheduler.set_timesteps with steps_offset=1 (the SD scheduler configs): arange(N)*(T//N) reversed + 1."""
    if kind == "dpm++2m":
        import numpy as np
        return [int(t) for t in np.linspace(0, num_train - 1, num_steps + 1).round()[::-1][:-1].astype(np.int64)]
    if kind == "ddim":
        ratio = num_train // num_steps
        return [i * ratio + 1 for i in range(num_steps)][::-1]
    raise ValueError(f"unknown sampler {kind!r} (dpm++2m | ddim)")


def step_coefficients(kind: str, ac: torch.Tensor, timesteps: Sequence[int], i: int):
    """(cx, c0, c1, alpha_t, sigma_t) of step i, all in f64 on the host.  DPM-Solver++ 2M with diffusers'
    defaults (solver_order 2, midpoint, lower_order_final for < 15 steps); DDIM eta=0, set_alpha_to_one=False."""
    ac = ac.double().cpu()
    t = timesteps[i]
    al, sg = ac.sqrt(), (1 - ac).sqrt()
    if kind == "ddim":
        ratio = ac.numel() // len(timesteps)
        tp = t - ratio
        a_t = ac[t]
        a_p = ac[tp] if tp >= 0 else ac[0]
        r = ((1 - a_p) / (1 - a_t)).sqrt()
        return float(r), float(a_p.sqrt() - r * a_t.sqrt()), 0.0, float(al[t]), float(sg[t])
    lam = al.log() - sg.log()
    n = len(timesteps)
    tp = 0 if i == n - 1 else timesteps[i + 1]
    h = lam[tp] - lam[t]
    cx = sg[tp] / sg[t]
    base = -al[tp] * (torch.exp(-h) - 1.0)
    if i == 0 or (i == n - 1 and n < 15):
        return float(cx), float(base), 0.0, float(al[t]), float(sg[t])
    r0 = (lam[t] - lam[timesteps[i - 1]]) / h
    return float(cx), float(base * (1 + 0.5 / r0)), float(-0.5 * base / r0), float(al[t]), float(sg[t])


class InferenceEngine:
    def __init__(self, cfg: sc.SDConfig, unet_w: Dict, vae_dec_w: Dict, clip_w: Dict, batch: int, height: int,
                 width: int, mapper_object: Dict[str, torch.Tensor], w_enc_object: torch.Tensor,
                 norm_scale_object: Optional[float], alpha_object: float = 0.2,
                 mapper_view: Optional[Dict[str, torch.Tensor]] = None, w_enc_view: Optional[torch.Tensor] = None,
                 norm_scale_view: Optional[float] = None, alpha_view: float = 0.2, n_view_params: int = 12,
                 unconstrained_object: bool = False, unconstrained_view: bool = False, hidden_object: int = 64,
                 device: str = "cuda", params_object: Optional[torch.Tensor] = None,
                 params_view: Optional[torch.Tensor] = None, object_slot: Optional[torch.Tensor] = None,
                 object_slot_stride: int = 0, legacy_pe_object: Optional[torch.Tensor] = None,
                 enc_dim_object: int = 64, output_bypass_object: bool = True, output_bypass_view: bool = True):
        """params_object / params_view: flat device buckets to ALIAS instead of copying the state dicts (validation
        during training reads the live parameters); object_slot (+stride) picks one mapper of a multi-object bucket."""
        self.cfg = cfg
        self.B = batch
        self.dev = device
        nlev = len(cfg.vae.block_out_channels)
        self.h, self.w = height >> (nlev - 1), width >> (nlev - 1)
        self.Lc = cfg.vae.latent_channels
        B, L, D = batch, cfg.clip.max_positions, cfg.clip.hidden_size
        self.L = L
        self.ac = alphas_cumprod(cfg.ddpm)
        # CFG-batched UNet: samples [0,B) unconditional, [B,2B) conditional
        self.unet = UNetEngine(cfg.unet, unet_w, 2 * batch, self.h, self.w, L, device, need_backward=False)
        nl = cfg.unet.n_cross_layers
        self.t_text = torch.zeros(B, dtype=torch.int64, device=device)
        self.ctx_k = torch.zeros((nl, B * L, D), dtype=lib.act_dtype(), device=device)
        self.ctx_v = torch.zeros_like(self.ctx_k)
        po = params_object if params_object is not None else flatten_mapper_state(mapper_object).to(device)
        mo = MapperState(po, w_enc_object.to(device).float().contiguous() if legacy_pe_object is None else None,
                         norm_scale_object, alpha_object, hidden=hidden_object, enc_dim=enc_dim_object,
                         unconstrained=unconstrained_object, slot=object_slot, slot_stride=object_slot_stride,
                         legacy_w_pe=(legacy_pe_object.to(device).float().contiguous()
                                      if legacy_pe_object is not None else None),
                         output_bypass=output_bypass_object)
        mv = None
        if mapper_view is not None or params_view is not None:
            pv = params_view if params_view is not None else flatten_mapper_state(mapper_view).to(device)
            mv = MapperState(pv, w_enc_view.to(device).float().contiguous(), norm_scale_view, alpha_view,
                             unconstrained=unconstrained_view, output_bypass=output_bypass_view)
        self.text = TextEngine(cfg.clip, clip_w, nl, batch, self.t_text, self.ctx_k, self.ctx_v, None, None, mo, None,
                               mv, None, n_view_params, False, device, need_backward=False)
        self.text.training = False
        # masks exist from the start: set_truncation() then mutates them in place and a captured sampler graph
        # (whose launches bake the mask pointer in) honours a truncation_idx set after the capture
        self.text.ensure_masks()
        self.decoder = VAEDecoderEngine(cfg.vae, vae_dec_w, batch, self.h, self.w, device)
        shape = (batch, self.Lc, self.h, self.w)
        self.x = torch.zeros(shape, dtype=torch.float32, device=device)
        self.m_prev = torch.zeros_like(self.x)
        self.image = self.decoder.image
        self.step_idx = torch.zeros(1, dtype=torch.int32, device=device)
        # device tables a captured sampler step reads (row = step_idx): {alpha_t, sigma_t, cx, c0, c1}, timestep
        self.coef_table = torch.zeros((cfg.ddpm.num_train_timesteps, 5), dtype=torch.float32, device=device)
        self.ts_table = torch.zeros((cfg.ddpm.num_train_timesteps,), dtype=torch.int64, device=device)
        self._graph = None
        self._graph_key = None

    # ------------------------------------------------------------------ conditioning
    def set_negative_prompt(self, input_ids: torch.Tensor):
        """`negative_prompt_embeds` (sd_pipeline_call.py:35-39): the plain text encoder on the negative prompt;
        used as K and V source of every cross-attention layer of the unconditional half."""
        B, L = self.B, self.L
        ids = input_ids.view(-1, L)
        if ids.shape[0] == 1:
            ids = ids.expand(B, L)
        none = torch.full((B,), -1, dtype=torch.int64)
        self.text.set_batch(ids, none, none if self.text.mv is not None else None, None)
        self.t_text.zero_()
        self.text.forward()
        # without a placeholder the bypass variant equals the plain one, and every layer sees the same embedding
        self.unet.ctx_k[:, : B * L].copy_(self.ctx_k)
        self.unet.ctx_v[:, : B * L].copy_(self.ctx_k)

    def set_prompt(self, input_ids, placeholder_object, placeholder_view=None, view_params=None,
                   truncation_idx: Optional[int] = None):
        self.text.set_batch(input_ids, placeholder_object, placeholder_view, view_params)
        self.text.set_truncation(truncation_idx)

    # ------------------------------------------------------------------ the loop
    @torch.no_grad()
    def generate(self, latents: torch.Tensor, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 kind: str = "dpm++2m", decode: bool = True, use_graph: bool = True):
        """latents: (B, 4, h, w) N(0,1) draw (`prepare_latents`, init_noise_sigma = 1 for both samplers).
        Returns the images f32 [B, H, W, 3] in [0,1] (the array `numpy_to_pil` receives) or the final latents."""
        if guidance_scale <= 1.0:
            raise ValueError("sd_pipeline_call only defines the classifier-free-guidance branch (guidance_scale > 1)")
        B, L = self.B, self.L
        ts = inference_timesteps(kind, num_inference_steps, self.cfg.ddpm.num_train_timesteps)
        self.x.copy_(latents)
        self.m_prev.zero_()
        self.unet.x_in[:B].copy_(self.x)
        self.unet.x_in[B:].copy_(self.x)
        vpred = self.cfg.ddpm.prediction_type == "v_prediction"
        if use_graph:
            # one captured sampler step, replayed T times: timesteps and step scalars come from device tables
            rows = [step_coefficients(kind, self.ac, ts, i) for i in range(len(ts))]
            self.coef_table[: len(ts)].copy_(torch.tensor([[a, s_, cx, c0, c1] for (cx, c0, c1, a, s_) in rows],
                                                          dtype=torch.float32))
            self.ts_table[: len(ts)].copy_(torch.tensor(ts, dtype=torch.int64))
            self.step_idx.zero_()
            key = (guidance_scale, vpred)
            if self._graph is None or self._graph_key != key:
                self._capture(guidance_scale, vpred)
                self._graph_key = key
                self.step_idx.zero_()
                self.x.copy_(latents)
                self.m_prev.zero_()
                self.unet.x_in[:B].copy_(self.x)
                self.unet.x_in[B:].copy_(self.x)
            for _ in ts:
                self._graph.replay()
        else:
            for i, t in enumerate(ts):
                self.t_text.fill_(t)
                self.unet.timesteps.fill_(t)
                self.text.forward()
                self.unet.ctx_k[:, B * L:].copy_(self.ctx_k)
                self.unet.ctx_v[:, B * L:].copy_(self.ctx_v)
                self.unet.forward()
                cx, c0, c1, a_t, s_t = step_coefficients(kind, self.ac, ts, i)
                ops.cfg_sampler_step(self.unet.pred, self.x, self.m_prev, self.unet.x_in, B, self.Lc,
                                     self.h * self.w, guidance_scale, a_t, s_t, cx, c0, c1, vpred)
        if not decode:
            return self.x
        self.decoder.z_in.copy_(self.x)
        self.decoder.forward()
        return self.image

    # ------------------------------------------------------------------ the reference's own prompt_embeds contract
    def load_contexts(self, embed) -> None:
        """Write ONE step's conditional conditioning into the UNet's per-layer K / V sources, as the 16 XTIAttenProc
        instances would read it (models/xti_attention_processor.py:27-41): a dict {CONTEXT_TENSOR_l ->  K source,
        CONTEXT_TENSOR_BYPASS_l -> V source (absent: the former)} or one tensor used by every layer for both."""
        B, L, nl = self.B, self.L, self.cfg.unet.n_cross_layers
        ck, cv = self.unet.ctx_k[:, B * L:], self.unet.ctx_v[:, B * L:]

        def rows(t):
            t = torch.as_tensor(t)
            if t.dim() == 2:
                t = t[None]
            if t.shape[0] == 1 and B > 1:
                t = t.expand(B, *t.shape[1:])
            if tuple(t.shape) != (B, L, ck.shape[-1]):
                raise ValueError(f"context of shape {tuple(t.shape)}: expected ({B}, {L}, {ck.shape[-1]})")
            return t.reshape(B * L, -1)

        if isinstance(embed, dict):
            for l in range(nl):
                k = embed[f"CONTEXT_TENSOR_{l}"]
                v = embed.get(f"CONTEXT_TENSOR_BYPASS_{l}", k)
                ck[l].copy_(rows(k))
                cv[l].copy_(rows(v))
        else:
            r = rows(embed)
            for l in range(nl):
                ck[l].copy_(r)
                cv[l].copy_(r)

    @torch.no_grad()
    def generate_from_contexts(self, latents: torch.Tensor, prompt_embeds, num_inference_steps: int = 50,
                               guidance_scale: float = 7.5, kind: str = "dpm++2m", decode: bool = True):
        """`sd_pipeline_call` with the conditioning ALREADY computed, exactly as the reference passes it
        (sd_pipeline_call.py:86: `prompt_embeds[i] if type(prompt_embeds) == list else prompt_embeds`): a list of T
        per-step XTI dicts (PromptManager.embed_prompt's return value, prompt_manager.py:79-99), one dict, or one tensor.
        The engine's own text pass is skipped; the negative prompt must have been set (set_negative_prompt)."""
        if guidance_scale <= 1.0:
            raise ValueError("sd_pipeline_call only defines the classifier-free-guidance branch (guidance_scale > 1)")
        B = self.B
        ts = inference_timesteps(kind, num_inference_steps, self.cfg.ddpm.num_train_timesteps)
        if type(prompt_embeds) == list and len(prompt_embeds) < len(ts):
            raise ValueError(f"{len(prompt_embeds)} per-step prompt embeddings for {len(ts)} sampler steps")
        self.x.copy_(latents)
        self.m_prev.zero_()
        self.unet.x_in[:B].copy_(self.x)
        self.unet.x_in[B:].copy_(self.x)
        vpred = self.cfg.ddpm.prediction_type == "v_prediction"
        if type(prompt_embeds) != list:
            self.load_contexts(prompt_embeds)
        for i, t in enumerate(ts):
            if type(prompt_embeds) == list:
                self.load_contexts(prompt_embeds[i])
            self.unet.timesteps.fill_(t)
            self.unet.forward()
            cx, c0, c1, a_t, s_t = step_coefficients(kind, self.ac, ts, i)
            ops.cfg_sampler_step(self.unet.pred, self.x, self.m_prev, self.unet.x_in, B, self.Lc, self.h * self.w,
                                 guidance_scale, a_t, s_t, cx, c0, c1, vpred)
        if not decode:
            return self.x
        self.decoder.z_in.copy_(self.x)
        self.decoder.forward()
        return self.image

    def _one_step(self, guidance_scale, vpred):
        B, L = self.B, self.L
        ops.table_fill_i64(self.t_text, self.ts_table, self.step_idx)
        ops.table_fill_i64(self.unet.timesteps, self.ts_table, self.step_idx)
        self.text.forward()
        self.unet.ctx_k[:, B * L:].copy_(self.ctx_k)
        self.unet.ctx_v[:, B * L:].copy_(self.ctx_v)
        self.unet.forward()
        ops.cfg_sampler_step_table(self.unet.pred, self.x, self.m_prev, self.unet.x_in, B, self.Lc, self.h * self.w,
                                   guidance_scale, self.coef_table, self.step_idx, vpred)
        ops.counter_advance(self.step_idx)

    def _capture(self, guidance_scale, vpred):
        """capture one sampler step (the warm-up run below is a real step: the caller re-seeds x afterwards)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._one_step(guidance_scale, vpred)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, stream=s):
                self._one_step(guidance_scale, vpred)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

    def memory_bytes(self) -> int:
        return self.unet.bytes + self.text.bytes + self.decoder.bytes
