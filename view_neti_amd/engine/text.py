"""NeTI text conditioning as one batched HIP launch schedule.

Reference: `Coach.get_text_conditioning` (training/coach.py:276-311) runs the text encoder once per
UNet cross-attention layer — 16 sequential passes, each with a mapper call, host syncs
(`.item()`, python `all(...)`) and ~180 tiny kernels.  Here the 16 passes are the batch dimension:
rows are ordered (layer, sample, token), so every CLIP linear is one M = 16*B*77 GEMM and the
outputs land directly in the UNet engine's per-layer key/value context buffers.

Numerics follow accelerate's fp16 autocast of the text encoder (training/coach.py:97-99,775-782):
fp32 embeddings and residual stream, fp16 matmul operands with fp32 accumulation, fp32 LayerNorm
statistics; the context is rounded to fp16 after the final LayerNorm (coach.py:299-304).

The backward is dgrad-only through CLIP (frozen, coach.py:646-653) down to the placeholder rows of
the embedding gradient, where the fused mapper backward produces the only weight gradients of the
whole train step.
"""
from __future__ import annotations

from functools import partial
from typing import Dict, Optional

import torch

from .. import ops
from .. import sd_config as sc
from .schedule import Schedule, rup


class MapperState:
    """Flat f32 parameter bucket of one NeTIMapper + its encoder frequencies: the Fourier `w_enc` of the paper's
    arch_view_net = 15, or (`legacy_w_pe` given) the 1024 x 2 frequencies of the legacy NeTIPositionalEncoding, in
    which case the bucket ends with the trainable input_layer [enc_dim][2*num_w] + bias and enc_dim = 10*16 = 160."""

    def __init__(self, params: torch.Tensor, w_enc: Optional[torch.Tensor], norm_scale: Optional[float], alpha: float,
                 hidden: int = 64, enc_dim: int = 64, unconstrained: bool = False, nested_dropout_prob: float = 0.0,
                 slot: Optional[torch.Tensor] = None, slot_stride: int = 0, legacy_w_pe: Optional[torch.Tensor] = None,
                 output_bypass: bool = True):
        # output_bypass False (models/neti_mapper.py:79-81,419-424): the mapper emits the word embedding only — no
        # CONTEXT_TENSOR_BYPASS, the value context equals the key context
        self.output_bypass = output_bypass
        self.legacy_w_pe = legacy_w_pe  # device f32 [num_w][2]; None = Fourier path
        self.pe_dim = 2 * legacy_w_pe.shape[0] if legacy_w_pe is not None else 0
        self.params = params            # flat bucket (of `slot_stride`-spaced mappers when slot is given)
        self.w_enc = w_enc
        self.norm_scale = norm_scale
        self.alpha = alpha
        self.hidden = hidden
        self.enc_dim = enc_dim
        self.unconstrained = unconstrained              # bypass_unconstrained (neti_mapper.py:130)
        self.nested_dropout_prob = nested_dropout_prob  # 0 disables (use_nested_dropout=False)
        self.slot = slot                # device int32[1]: which mapper of the bucket (mapper_object_lookup)
        self.slot_stride = slot_stride


def flatten_mapper_state(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """state_dict (reference key names, checkpoint_handler.py:57-97) -> flat bucket in the order the
    kernels expect; the legacy mapper's input_layer (neti_mapper.py:155-163) goes LAST."""
    keys = ["net.0.weight", "net.0.bias", "net.1.weight", "net.1.bias", "net.3.weight", "net.3.bias",
            "net.4.weight", "net.4.bias", "output_layer.0.weight", "output_layer.0.bias"]
    if "input_layer.weight" in sd:
        keys += ["input_layer.weight", "input_layer.bias"]
    return torch.cat([sd[k].reshape(-1).float() for k in keys])


def unflatten_mapper_state(flat: torch.Tensor, enc_dim: int, hidden: int, out_dim: int, pe_dim: int = 0
                           ) -> Dict[str, torch.Tensor]:
    shapes = [("net.0.weight", (hidden, enc_dim)), ("net.0.bias", (hidden,)), ("net.1.weight", (hidden,)),
              ("net.1.bias", (hidden,)), ("net.3.weight", (hidden, hidden)), ("net.3.bias", (hidden,)),
              ("net.4.weight", (hidden,)), ("net.4.bias", (hidden,)), ("output_layer.0.weight", (out_dim, hidden)),
              ("output_layer.0.bias", (out_dim,))]
    if pe_dim:
        shapes += [("input_layer.weight", (enc_dim, pe_dim)), ("input_layer.bias", (enc_dim,))]
    out, o = {}, 0
    for k, shp in shapes:
        n = 1
        for s in shp:
            n *= s
        out[k] = flat[o:o + n].reshape(shp).clone()
        o += n
    assert o == flat.numel()
    return out


class TextEngine(Schedule):
    def __init__(self, cfg: sc.CLIPTextConfig, weights: Dict[str, torch.Tensor], n_layers: int, batch: int,
                 timesteps: torch.Tensor, ctx_k: torch.Tensor, ctx_v: torch.Tensor, dctx_k: torch.Tensor,
                 dctx_v: torch.Tensor, mapper_object: MapperState, grads_object: torch.Tensor,
                 mapper_view: Optional[MapperState] = None, grads_view: Optional[torch.Tensor] = None,
                 n_view_params: int = 12, train_view: bool = True, device: str = "cuda",
                 need_backward: bool = True, autotune: bool = True, rng_state: Optional[torch.Tensor] = None):
        super().__init__(batch, 32, cfg.eps, device, need_backward)
        self.cfg = cfg
        self.nl = n_layers
        self.L = cfg.max_positions
        B, L, D, nl = batch, self.L, cfg.hidden_size, n_layers
        self.R = nl * B
        self.Rt = nl * B * L
        self.timesteps = timesteps
        self.ctx_k, self.ctx_v, self.dctx_k, self.dctx_v = ctx_k, ctx_v, dctx_k, dctx_v
        assert ctx_k.numel() == self.Rt * D and ctx_k.is_contiguous()
        self.mo, self.mv = mapper_object, mapper_view
        self.go, self.gv = grads_object, grads_view
        self.train_view = train_view and mapper_view is not None
        # per-batch inputs (static buffers, refreshed by set_batch)
        self.ids = torch.zeros((B, L), dtype=torch.int64, device=device)
        self.pos_obj = torch.zeros((B,), dtype=torch.int32, device=device)
        self.rows_obj = torch.zeros((self.R,), dtype=torch.int32, device=device)
        self.pos_view = torch.full((B,), -1, dtype=torch.int32, device=device) if mapper_view else None
        self.rows_view = torch.zeros((self.R,), dtype=torch.int32, device=device) if mapper_view else None
        self.view_params = self._buf((B, n_view_params), torch.float32, zero=True) if mapper_view else None
        # nested dropout (neti_mapper.py:401-414): 0/1 masks over the mappers' hidden vectors, redrawn every
        # step from the device RNG while `training`; None = feature off
        self.rng_state = rng_state
        self.training = True
        self.hidden_mask_obj = self._mask_buf(mapper_object)
        self.hidden_mask_view = self._mask_buf(mapper_view) if mapper_view is not None else None
        self.norm_terms = self._buf((2, self.R), torch.float32, zero=True)
        self.accumulate_grads = False  # True on micro-steps 2..k of a gradient-accumulation group
        self.tok_emb = self._w32(weights["text_model.embeddings.token_embedding.weight"])
        self.pos_emb = self._w32(weights["text_model.embeddings.position_embedding.weight"])
        self._build(weights)
        if need_backward:
            self._build_backward()
        if autotune:
            self.autotune()
        self.bind_workspace()

    def _mask_buf(self, m: MapperState):
        if m.nested_dropout_prob <= 0.0:
            return None
        if self.rng_state is None:
            raise ValueError("nested dropout needs the step engine's device RNG state")
        return torch.ones((self.R, m.hidden), dtype=torch.float32, device=self.dev)

    def _draw_masks(self):
        for mask, m, sid in ((self.hidden_mask_obj, self.mo, 3), (self.hidden_mask_view, self.mv, 4)):
            if mask is None:
                continue
            if self.training:
                ops.nested_dropout_mask(mask, self.nl, self.B, m.hidden, m.nested_dropout_prob, self.rng_state, sid)
            else:
                mask.fill_(1.0)  # eval: no dropout (truncation_idx is an inference-only knob)

    def ensure_masks(self):
        """allocate all-ones hidden masks where nested dropout is off, so that `set_truncation` only ever mutates
        buffers IN PLACE: a sampler graph captured before the first truncated call then still reads the mask (a
        lazily created tensor would leave the captured launches with a null mask pointer)."""
        for name, m in (("hidden_mask_obj", self.mo), ("hidden_mask_view", self.mv)):
            if m is not None and getattr(self, name) is None:
                setattr(self, name, torch.ones((self.R, m.hidden), dtype=torch.float32, device=self.dev))

    def set_truncation(self, truncation_idx: Optional[int]):
        """inference-time truncation of the mapper's hidden vector (`truncation_idx`, neti_mapper.py:409-411):
        hidden[idx:] = 0 for every call; None restores the full vector."""
        for name, m in (("hidden_mask_obj", self.mo), ("hidden_mask_view", self.mv)):
            if m is None:
                continue
            mask = getattr(self, name)
            if truncation_idx is None:
                if mask is not None:
                    mask.fill_(1.0)
                continue
            if mask is None:
                mask = torch.ones((self.R, m.hidden), dtype=torch.float32, device=self.dev)
                setattr(self, name, mask)
            mask.fill_(1.0)
            mask[:, truncation_idx:] = 0.0

    # ------------------------------------------------------------------ batch plumbing
    def set_batch(self, input_ids: torch.Tensor, placeholder_object: torch.Tensor,
                  placeholder_view: Optional[torch.Tensor] = None, view_params: Optional[torch.Tensor] = None, upload=None):
        """Host-side equivalent of the `locs = (input_ids == placeholder)` bookkeeping of
        models/net_clip_text_embedding.py:95-97,127-129 (one placeholder per row, asserted).
        upload(name, device_tensor, host_tensor): how the per-batch index buffers reach the device (the train step passes its
        pinned stager, engine/staging.py; default: a plain blocking copy_)."""
        if upload is None:
            upload = lambda name, dev, host: dev.copy_(host)
        B, L, nl = self.B, self.L, self.nl
        ids = input_ids.cpu()
        assert tuple(ids.shape) == (B, L)

        def positions(ph):
            ph = ph.cpu().view(B, 1)
            locs = ids == ph
            if not bool((locs.sum(1) == 1).all()):
                raise ValueError("each prompt must contain its placeholder token exactly once")
            return locs.float().argmax(1).to(torch.int32)

        if bool((placeholder_object.cpu() == -1).all()):
            # no learned object token in the prompt (mode 1 captions, the negative prompt of inference):
            # position -1 never matches, so the rows keep their vocabulary embedding and there is no bypass
            po = torch.full((B,), -1, dtype=torch.int32)
        else:
            po = positions(placeholder_object)
        upload("ids", self.ids, ids)
        upload("pos_obj", self.pos_obj, po)
        base = (torch.arange(nl).view(nl, 1) * B + torch.arange(B).view(1, B)) * L
        # rows of the residual stream whose dX the mapper backward gathers; -1 (no placeholder in the prompt) = no gradient:
        # the mapper's output reached nothing, its bucket segment must stay frozen (vneti_mapper_bwd zeroes such rows)
        upload("rows_obj", self.rows_obj, torch.where(po.view(1, B) >= 0, base + po.view(1, B), torch.full_like(base, -1))
               .reshape(-1).to(torch.int32))
        if self.mv is not None:
            if placeholder_view is None or bool((placeholder_view.cpu() == -1).all()):
                self.pos_view.fill_(-1)
            else:
                pv = positions(placeholder_view)
                upload("pos_view", self.pos_view, pv)
                upload("rows_view", self.rows_view, (base + pv.view(1, B)).reshape(-1).to(torch.int32))
                upload("view_params", self.view_params, view_params.float())

    # ------------------------------------------------------------------ schedule
    def _mapper_bufs(self, m: MapperState, nfeat):
        D, R = self.cfg.hidden_size, self.R
        return dict(data=self._buf((R, nfeat), torch.float32), word=self._buf((R, D), torch.float32),
                    byp=self._buf((R, D), torch.float32), dbyp=self._buf((R, D), torch.float32, zero=True),
                    save=self._buf((ops.mapper_save_floats(R, m.enc_dim, m.hidden),), torch.float32),
                    rowg=self._buf((ops.mapper_rowgrad_floats(R, m.hidden, D, m.output_bypass),), torch.float32))

    def _build(self, w):
        cfg = self.cfg
        B, L, D, nl, R, Rt = self.B, self.L, cfg.hidden_size, self.nl, self.R, self.Rt
        H = cfg.num_heads
        hd = D // H
        F = cfg.intermediate_size
        f = self.fwd
        mo = self.mo
        self.bo = self._mapper_bufs(mo, 2)
        if self.hidden_mask_obj is not None or self.hidden_mask_view is not None:
            f.append(self._draw_masks)
        if mo.legacy_w_pe is not None:
            # legacy object mapper (arch_view_net <= 14): NeTIPositionalEncoding of the raw (t, l) -> trainable input_layer
            # (the tail of the bucket) -> the same MLP kernels fed through `enc_in`
            self.n_std_obj = ops.mapper_num_params(mo.enc_dim, mo.hidden, D, mo.output_bypass)
            self.bo["enc"] = self._buf((R, mo.enc_dim), torch.float32)
            self.bo["denc"] = self._buf((R, mo.enc_dim), torch.float32)
            f.append(lambda: ops.mapper_legacy_input_fwd(mo.params[self.n_std_obj:], self.timesteps, mo.legacy_w_pe,
                                                         self.bo["enc"], nl, B, mo.enc_dim, mo.pe_dim, mo.slot,
                                                         mo.slot_stride))
            f.append(lambda: ops.mapper_fwd(mo.params, None, None, self.hidden_mask_obj, mo.norm_scale, self.bo["word"],
                                            self.bo["byp"], self.bo["save"], R, mo.enc_dim, mo.hidden, D, mo.output_bypass,
                                            mo.slot, mo.slot_stride, enc_in=self.bo["enc"]))
        else:
            f.append(partial(ops.mapper_inputs, self.timesteps, None, self.bo["data"], nl, B))
            f.append(lambda: ops.mapper_fwd(mo.params, self.bo["data"], mo.w_enc, self.hidden_mask_obj, mo.norm_scale,
                                            self.bo["word"], self.bo["byp"], self.bo["save"], R, mo.enc_dim, mo.hidden,
                                            D, mo.output_bypass, mo.slot, mo.slot_stride))
        self.bv = None
        if self.mv is not None:
            mv = self.mv
            self.bv = self._mapper_bufs(mv, 2 + self.view_params.shape[1])
            f.append(partial(ops.mapper_inputs, self.timesteps, self.view_params, self.bv["data"], nl, B))
            f.append(lambda: ops.mapper_fwd(mv.params, self.bv["data"], mv.w_enc, self.hidden_mask_view,
                                            mv.norm_scale, self.bv["word"], self.bv["byp"], self.bv["save"], R,
                                            mv.enc_dim, mv.hidden, D, mv.output_bypass))
        x = self._buf((Rt, D), torch.float32)
        f.append(partial(ops.text_embed, self.tok_emb, self.pos_emb, self.ids, self.pos_obj, self.bo["word"],
                         self.pos_view, self.bv["word"] if self.bv else None, x, nl, B, L, D))
        self.x0 = x
        self.layers = []
        act = ops.ACT_QUICK_GELU if cfg.act == "quick_gelu" else ops.ACT_GELU
        for i in range(cfg.num_layers):
            p = f"text_model.encoder.layers.{i}."
            r = dict(x_in=x, act=act)
            n1, r["ln1"] = self._ln(x, p + "layer_norm1", w)
            wqkv = torch.cat([w[p + "self_attn.q_proj.weight"], w[p + "self_attn.k_proj.weight"],
                              w[p + "self_attn.v_proj.weight"]], 0)
            bqkv = torch.cat([w[p + "self_attn.q_proj.bias"], w[p + "self_attn.k_proj.bias"],
                              w[p + "self_attn.v_proj.bias"]], 0)
            r["wqkv"], r_bqkv = self._w16(wqkv), self._w32(bqkv)
            qkv = self._buf((Rt, 3 * D))
            f.append(partial(ops.gemm, n1, r["wqkv"], qkv, bias=r_bqkv))
            q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            o = self._buf((Rt, D))
            lse = self._buf((R, H, L), torch.float32)
            f.append(partial(ops.attn_fwd, q, k, v, o, lse, R, H, L, L, hd, hd ** -0.5, True))
            r["wo"], bo_ = self._w16(w[p + "self_attn.out_proj.weight"]), self._w32(w[p + "self_attn.out_proj.bias"])
            x_mid = self._buf((Rt, D), torch.float32)
            f.append(partial(ops.gemm, o, r["wo"], x_mid, bias=bo_, resid=x))
            n2, r["ln2"] = self._ln(x_mid, p + "layer_norm2", w)
            r["w1"], b1 = self._w16(w[p + "mlp.fc1.weight"]), self._w32(w[p + "mlp.fc1.bias"])
            r["w2"], b2 = self._w16(w[p + "mlp.fc2.weight"]), self._w32(w[p + "mlp.fc2.bias"])
            f1 = self._buf((Rt, F))
            a1 = self._buf((Rt, F))
            # fc1 writes the pre-activation (kept for backward) and the activated tensor in one epilogue
            f.append(partial(ops.gemm, n2, r["w1"], f1, bias=b1, out2=a1, act2=act))
            x_out = self._buf((Rt, D), torch.float32)
            f.append(partial(ops.gemm, a1, r["w2"], x_out, bias=b2, resid=x_mid))
            r.update(qkv=qkv, o=o, lse=lse, x_mid=x_mid, f1=f1)
            if self.need_backward:
                tr = lambda t: self._w16(t.t())
                r["w2d"], r["w1d"] = tr(w[p + "mlp.fc2.weight"]), tr(w[p + "mlp.fc1.weight"])
                r["wod"], r["wqkvd"] = tr(w[p + "self_attn.out_proj.weight"]), tr(wqkv)
            self.layers.append(r)
            x = x_out
        self.last = x
        self.fln_g = self._w32(w["text_model.final_layer_norm.weight"])
        self.fln_b = self._w32(w["text_model.final_layer_norm.bias"])
        byp_o = self.bo["byp"] if self.mo.output_bypass else None   # None: no bypass row for that mapper, ctx_v = ctx_k there
        byp_v = self.bv["byp"] if (self.bv and self.mv.output_bypass) else None
        f.append(lambda: ops.text_final_fwd(self.last, self.fln_g, self.fln_b, cfg.eps, self.pos_obj, byp_o,
                                            self.mo.alpha, self.pos_view, byp_v,
                                            self.mv.alpha if self.mv else 0.0, self.ctx_k, self.ctx_v, nl, B, L, D,
                                            self.mo.unconstrained, bool(self.mv and self.mv.unconstrained),
                                            self.norm_terms))

    def _build_backward(self):
        cfg = self.cfg
        B, L, D, nl, R, Rt = self.B, self.L, cfg.hidden_size, self.nl, self.R, self.Rt
        H = cfg.num_heads
        hd = D // H
        F = cfg.intermediate_size
        bw = self.bwd
        dx = self._buf((Rt, D), torch.float32)
        byp_o = self.bo["byp"] if self.mo.output_bypass else None
        byp_v = self.bv["byp"] if (self.bv and self.mv.output_bypass) else None
        bw.append(lambda: ops.text_final_bwd(self.last, self.fln_g, cfg.eps, self.pos_obj, byp_o,
                                             self.mo.alpha, self.bo["dbyp"] if byp_o is not None else None, self.pos_view,
                                             byp_v, self.mv.alpha if self.mv else 0.0,
                                             self.bv["dbyp"] if byp_v is not None else None, self.dctx_k, self.dctx_v, dx, nl, B,
                                             L, D, self.mo.unconstrained, bool(self.mv and self.mv.unconstrained),
                                             self.norm_terms))
        g16 = self._buf((Rt, D))
        dxm = self._buf((Rt, D), torch.float32)
        # the residual-stream gradient is f32; the dgrad GEMMs want f16 operands: every LayerNorm backward
        # writes that f16 copy itself, only the very first one (out of text_final_bwd) needs a cast launch
        bw.append(partial(ops.cast_f32_f16, dx, g16))
        for r in reversed(self.layers):
            qkv = r["qkv"]
            q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
            df1 = self._tmp("cB", Rt, F)   # (dx_out @ W2) * act'(f1): activation backward in the dgrad epilogue
            bw.append(partial(ops.gemm, g16, r["w2d"], df1, gate=r["f1"], gate_act=r["act"]))
            dn2 = self._tmp("cC", Rt, D)
            bw.append(partial(ops.gemm, df1, r["w1d"], dn2))
            bw.append(partial(self._ln_bwd, r["ln2"], dn2, dxm, dx, g16))     # dx_mid = LN2'(dn2) + dx_out
            do = self._tmp("cD", Rt, D)
            bw.append(partial(ops.gemm, g16, r["wod"], do))
            delta = self._tmp("cdelta", R * H, L, torch.float32)
            dqkv = self._tmp("cE", Rt, 3 * D)
            dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
            sc_ = hd ** -0.5
            if ops.attn_bwd_small_ok(L, hd):  # 77 tokens, 64-wide heads: dQ, dK, dV of a (sequence, head) in one launch
                bw.append(partial(ops.attn_bwd_small, q, k, v, do, r["o"], r["lse"], dq, dk, dv, R, H, L, hd, sc_, True))
            else:
                # dQ first: it also produces delta = rowsum(dO o O) for the dK/dV kernel
                bw.append(partial(ops.attn_bwd_dq, q, k, v, do, r["lse"], delta, dq, R, H, L, L, hd, sc_, True, O=r["o"]))
                bw.append(partial(ops.attn_bwd_dkv, q, k, v, do, r["lse"], delta, dk, dv, R, H, L, L, hd, sc_, True))
            dn1 = self._tmp("cC", Rt, D)
            bw.append(partial(ops.gemm, dqkv, r["wqkvd"], dn1))
            bw.append(partial(self._ln_bwd, r["ln1"], dn1, dx, dxm, g16))     # dx_in = LN1'(dn1) + dx_mid
        self.dx0 = dx
        mo = self.mo
        bw.append(lambda: ops.mapper_bwd(mo.params, self.hidden_mask_obj, mo.norm_scale, self.bo["word"], self.dx0,
                                         self.rows_obj, D, self.bo["dbyp"] if mo.output_bypass else None, self.bo["save"],
                                         self.bo["rowg"], self.go, self.accumulate_grads, R, mo.enc_dim, mo.hidden, D,
                                         mo.output_bypass, mo.slot, mo.slot_stride, denc=self.bo.get("denc")))
        if mo.legacy_w_pe is not None:
            bw.append(lambda: ops.mapper_legacy_input_bwd(self.timesteps, mo.legacy_w_pe, self.bo["denc"],
                                                          self.go[self.n_std_obj:], self.accumulate_grads, self.nl, self.B,
                                                          mo.enc_dim, mo.pe_dim, mo.slot, mo.slot_stride))
        if self.train_view:
            mv = self.mv
            bw.append(lambda: ops.mapper_bwd(mv.params, self.hidden_mask_view, mv.norm_scale, self.bv["word"], self.dx0,
                                             self.rows_view, D, self.bv["dbyp"] if mv.output_bypass else None,
                                             self.bv["save"], self.bv["rowg"], self.gv, self.accumulate_grads, R, mv.enc_dim,
                                             mv.hidden, D, mv.output_bypass))
