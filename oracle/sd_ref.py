"""CPU ORACLE (test infrastructure, never the product path).

Plain-PyTorch fp32 restatement of the graph the ViewNeTI train step runs
(training/coach.py:151-264).  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; the shipped path (view_neti_amd/) never does.

Two kinds of functions live here:
  * restatements of REFERENCE-OWNED code, each citing the file:line it follows —
    NeTIMapper (arch_view_net=15), FourierPositionalEncodingNDims, NeTICLIPTextEmbeddings,
    the textual-bypass injection of NeTICLIPTextTransformer, XTIAttenProc, Coach.get_text_conditioning
    and the loss of Coach.train.  These are pinned against the real reference modules imported
    in the build container (oracle/make_golden.py -> tests/golden/*.npz).
  * restatements of THIRD-PARTY graphs that the reference reaches through diffusers 0.14 /
    transformers 4.27.4 (UNet2DConditionModel, AutoencoderKL.encode, CLIPEncoder, DDPMScheduler).
    Those packages are absent from /root/reference and from this image; the published
    architecture is restated from SURVEY.md Appendix A.  The CLIP encoder stack is pinned
    against transformers' own CLIPTextModel (tests/golden/clip_tiny.npz); the diffusers graphs
    have no available implementation to pin against: PARITY UNPINNED for UNet/VAE/DDPM.

Everything is functional: weights are dicts of fp32 tensors keyed by the diffusers/transformers
state-dict names (view_neti_amd/sd_config.py enumerates them).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------
# reference-owned: positional encoding + NeTI mapper (arch_view_net = 15)
# ------------------------------------------------------------------------------------------
def fourier_w(sigmas, dim: int = 64, seed: int = 0) -> torch.Tensor:
    """models/positional_encoding.py:154-171 — w ~ N(0,1) drawn right after
    torch.manual_seed(seed) (global RNG side effect), column i scaled by
    sigma_i.  Returned without touching the caller's RNG state."""
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    w = torch.randn((dim // 2, len(sigmas)))
    torch.random.set_rng_state(state)
    for i, s in enumerate(sigmas):
        w[:, i] *= s
    return w


def fourier_encode(w: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """models/positional_encoding.py:174-195: v = cat[sin(w x^T), cos(w x^T)]^T  -> (bs, dim)"""
    if x.ndim == 1:
        x = x.unsqueeze(1)
    proj = w @ x.t()
    return torch.cat([torch.sin(proj), torch.cos(proj)]).t()


def mapper_forward(p: W, w_enc: torch.Tensor, timestep: torch.Tensor, unet_layer: torch.Tensor,
                   norm_scale: Optional[float], output_bypass: bool = True, num_unet_layers: int = 16,
                   view_params: Optional[torch.Tensor] = None, truncation_mask: Optional[torch.Tensor] = None):
    """models/neti_mapper.py:165-197 with arch_view_net=15:
    do_positional_encoding (:542-578) -> net (:148-152 / :603-607) -> [nested dropout :401-414,
    expressed as a 0/1 mask on the hidden vector] -> output_layer (:153/:608) -> get_output
    (:416-438).  `view_params` are the already scaled [-1,1] camera parameters (:294-337).
    p keys: net.0/1/3/4.{weight,bias}, output_layer.0.{weight,bias}."""
    t = timestep.float() / 1000 * 2 - 1
    l = unet_layer.float() / num_unet_layers * 2 - 1
    data = torch.stack((t, l), dim=1)
    if view_params is not None:
        data = torch.cat((data, view_params), dim=1)
    enc = fourier_encode(w_enc, data)
    h = F.linear(enc, p["net.0.weight"], p["net.0.bias"])
    h = F.leaky_relu(F.layer_norm(h, (h.shape[-1],), p["net.1.weight"], p["net.1.bias"]))
    h = F.linear(h, p["net.3.weight"], p["net.3.bias"])
    h = F.leaky_relu(F.layer_norm(h, (h.shape[-1],), p["net.4.weight"], p["net.4.bias"]))
    if truncation_mask is not None:
        h = h * truncation_mask
    out = F.linear(h, p["output_layer.0.weight"], p["output_layer.0.bias"])
    if output_bypass:
        dim = out.shape[1] // 2
        word, bypass = out[:, :dim], out[:, dim:]
    else:
        word, bypass = out, None
    if norm_scale is not None:
        word = F.normalize(word, dim=-1) * norm_scale
    return word, bypass


def neti_pe_encode(w: torch.Tensor, t: torch.Tensor, l: torch.Tensor) -> torch.Tensor:
    """models/positional_encoding.py:23-41 (legacy `NeTIPositionalEncoding.encode`, batched branch): x = (t, l) RAW
    (timestep 0..999, layer index 0..15 — no [-1,1] scaling), v = cat[sin(w x), cos(w x)] over num_w = 1024
    frequencies, L2-normalised per sample -> (bs, 2048).  (|v| = sqrt(num_w) exactly, sin^2 + cos^2 = 1.)"""
    x = torch.stack([t.float(), l.float()], dim=1).t()
    v = torch.cat([torch.sin(w @ x), torch.cos(w @ x)])
    return (v / v.norm(dim=0)).t()


def neti_pe_init_layer(w: torch.Tensor, num_time_anchors: int = 10, num_layers: int = 16) -> torch.Tensor:
    """models/positional_encoding.py:43-51: the (anchors*layers, 2048) initial weight of the legacy `input_layer` —
    row (i*num_layers + j) is the normalised encoding of the anchor (t = i * (1000 // anchors), l = j)."""
    rows = []
    for t_anchor in range(0, 1000, 1000 // num_time_anchors):
        for l_anchor in range(num_layers):
            x = torch.tensor([float(t_anchor), float(l_anchor)])
            v = torch.cat([torch.sin(w @ x), torch.cos(w @ x)])
            rows.append(v / v.norm())
    return torch.stack(rows)


def mapper_forward_legacy(p: W, w_pe: torch.Tensor, timestep: torch.Tensor, unet_layer: torch.Tensor,
                          norm_scale: Optional[float], output_bypass: bool = True,
                          truncation_mask: Optional[torch.Tensor] = None):
    """models/neti_mapper.py:165-206,369-374 with arch_view_net <= 14 (the dataclass default 0, training/config.py:130),
    embedding_type 'object', use_positional_encoding 1: encode (above) -> input_layer Linear(2048, anchors*layers)
    (:155-163) -> net (Linear-LN-LeakyReLU x2, h = arch_mlp_hidden_dims, :148-152) -> [nested dropout mask] ->
    output_layer -> get_output (:416-438).
    p keys: input_layer.{weight,bias}, net.0/1/3/4.{weight,bias}, output_layer.0.{weight,bias}."""
    enc = neti_pe_encode(w_pe, timestep, unet_layer)
    e = F.linear(enc, p["input_layer.weight"], p["input_layer.bias"])
    h = F.linear(e, p["net.0.weight"], p["net.0.bias"])
    h = F.leaky_relu(F.layer_norm(h, (h.shape[-1],), p["net.1.weight"], p["net.1.bias"]))
    h = F.linear(h, p["net.3.weight"], p["net.3.bias"])
    h = F.leaky_relu(F.layer_norm(h, (h.shape[-1],), p["net.4.weight"], p["net.4.bias"]))
    if truncation_mask is not None:
        h = h * truncation_mask
    out = F.linear(h, p["output_layer.0.weight"], p["output_layer.0.bias"])
    if output_bypass:
        dim = out.shape[1] // 2
        word, bypass = out[:, :dim], out[:, dim:]
    else:
        word, bypass = out, None
    if norm_scale is not None:
        word = F.normalize(word, dim=-1) * norm_scale
    return word, bypass


# ------------------------------------------------------------------------------------------
# third-party: CLIP text encoder stack (transformers 4.27.4 CLIPEncoder), SURVEY App. A.2
# ------------------------------------------------------------------------------------------
def _act(x, kind):
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    return F.gelu(x)


def clip_encoder(w: W, cfg, x: torch.Tensor) -> torch.Tensor:
    """x: (B, L, D) embeddings -> last hidden state BEFORE final_layer_norm."""
    B, L, D = x.shape
    H = cfg.num_heads
    hd = D // H
    mask = torch.full((L, L), torch.finfo(x.dtype).min).triu(1)
    for i in range(cfg.num_layers):
        p = f"text_model.encoder.layers.{i}."
        r = x
        h = F.layer_norm(x, (D,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], cfg.eps)
        q = F.linear(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]) * hd ** -0.5
        k = F.linear(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"])
        v = F.linear(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"])
        q = q.view(B, L, H, hd).transpose(1, 2)
        k = k.view(B, L, H, hd).transpose(1, 2)
        v = v.view(B, L, H, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, -1)
        o = (a @ v).transpose(1, 2).reshape(B, L, D)
        x = r + F.linear(o, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"])
        r = x
        h = F.layer_norm(x, (D,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], cfg.eps)
        h = _act(F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"]), cfg.act)
        x = r + F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
    return x


# ------------------------------------------------------------------------------------------
# reference-owned: NeTI text embeddings + textual bypass
# ------------------------------------------------------------------------------------------
def neti_embeddings(token_emb: torch.Tensor, pos_emb: torch.Tensor, input_ids: torch.Tensor,
                    placeholder_object: Optional[torch.Tensor], word_object: Optional[torch.Tensor],
                    placeholder_view: Optional[torch.Tensor] = None, word_view: Optional[torch.Tensor] = None):
    """models/net_clip_text_embedding.py:34-137: E[ids]; rows whose id == the sample's placeholder
    id are overwritten with the mapper word embedding (object :95-98, view :127-130); + P[:L]."""
    x = token_emb[input_ids]  # (B, L, D) — a copy, like nn.Embedding's output
    B = input_ids.shape[0]
    if word_object is not None:
        locs = input_ids == placeholder_object.unsqueeze(1)
        assert bool((locs.sum(1) == 1).all())
        pos = locs.float().argmax(1)
        x = x.clone()
        x[torch.arange(B), pos] = word_object
    if word_view is not None:
        locs = input_ids == placeholder_view.unsqueeze(1)
        assert bool((locs.sum(1) == 1).all())
        pos = locs.float().argmax(1)
        x = x.clone()
        x[torch.arange(B), pos] = word_view
    return x + pos_emb[: input_ids.shape[1]].unsqueeze(0)


def apply_bypass(last_hidden: torch.Tensor, input_ids: torch.Tensor, placeholder: torch.Tensor,
                 bypass: torch.Tensor, unconstrained: bool, alpha: float) -> torch.Tensor:
    """models/neti_clip_text_encoder.py:129-153 (object) / :155-180 (view), applied to a clone of
    the last hidden state (:121)."""
    B = last_hidden.shape[0]
    idx = (input_ids == placeholder.unsqueeze(1)).float().argmax(1)
    out = last_hidden.clone()
    existing = out[torch.arange(B), idx]
    if not unconstrained:
        b = bypass / bypass.norm(dim=1, keepdim=True) * existing.norm(dim=1, keepdim=True)
        new = existing + alpha * b
    else:
        norm_term = out.norm(dim=-1).mean(-1).detach()
        new = bypass / bypass.norm(dim=1, keepdim=True) * norm_term.unsqueeze(1)
    out[torch.arange(B), idx] = new
    return out


def neti_text_encoder(w: W, cfg, input_ids, placeholder_object, word_object, bypass_object,
                      unconstrained_object=False, alpha_object=0.2, placeholder_view=None, word_view=None,
                      bypass_view=None, unconstrained_view=False, alpha_view=0.2):
    """NeTICLIPTextTransformer.forward (models/neti_clip_text_encoder.py:57-225), `batch=` branch.
    Returns (last_hidden_state, last_hidden_state_with_bypass | None), both after final LN."""
    x = neti_embeddings(w["text_model.embeddings.token_embedding.weight"],
                        w["text_model.embeddings.position_embedding.weight"], input_ids,
                        placeholder_object, word_object, placeholder_view, word_view)
    last = clip_encoder(w, cfg, x)
    lnw, lnb = w["text_model.final_layer_norm.weight"], w["text_model.final_layer_norm.bias"]
    D = last.shape[-1]
    with_bypass = None
    if bypass_object is not None or bypass_view is not None:
        wb = last
        if bypass_object is not None:
            wb = apply_bypass(wb, input_ids, placeholder_object, bypass_object, unconstrained_object, alpha_object)
        if bypass_view is not None:
            wb = apply_bypass(wb, input_ids, placeholder_view, bypass_view, unconstrained_view, alpha_view)
        with_bypass = F.layer_norm(wb, (D,), lnw, lnb, cfg.eps)
    return F.layer_norm(last, (D,), lnw, lnb, cfg.eps), with_bypass


def clip_plain(w: W, cfg, input_ids: torch.Tensor) -> torch.Tensor:
    """the plain `text_encoder(input_ids=...)[0]` path (NeTICLIPTextTransformer.forward without a
    NeTIBatch; sd_pipeline_call.py:35-39 uses it for the negative prompt): embeddings -> encoder ->
    final_layer_norm."""
    x = neti_embeddings(w["text_model.embeddings.token_embedding.weight"],
                        w["text_model.embeddings.position_embedding.weight"], input_ids, None, None)
    last = clip_encoder(w, cfg, x)
    return F.layer_norm(last, (last.shape[-1],), w["text_model.final_layer_norm.weight"],
                        w["text_model.final_layer_norm.bias"], cfg.eps)


# ------------------------------------------------------------------------------------------
# reference-owned: XTI attention processor
# ------------------------------------------------------------------------------------------
def xti_attention(wq, wk, wv, wo, bo, heads: int, hidden: torch.Tensor, ehs):
    """models/xti_attention_processor.py:9-57.  `ehs` is None (self-attention), a tensor, or the
    context dict {"this_idx", "CONTEXT_TENSOR_i", "CONTEXT_TENSOR_BYPASS_i"}: K from the former,
    V from the latter, counter advanced mod 16 (:16-22)."""
    ctx, ctx_bypass = None, None
    if ehs is not None:
        if isinstance(ehs, dict):
            i = ehs["this_idx"]
            ctx = ehs[f"CONTEXT_TENSOR_{i}"]
            ctx_bypass = ehs.get(f"CONTEXT_TENSOR_BYPASS_{i}")
            ehs["this_idx"] = (i + 1) % 16
        else:
            ctx = ehs
    q = F.linear(hidden, wq)
    if ctx is None:
        ctx = hidden
    k = F.linear(ctx, wk)
    v = F.linear(ctx_bypass if ctx_bypass is not None else ctx, wv)
    B, N, C = q.shape
    d = C // heads

    def split(t):
        return t.view(B, t.shape[1], heads, d).transpose(1, 2)

    probs = torch.softmax(split(q) @ split(k).transpose(-1, -2) * d ** -0.5, -1)
    o = (probs @ split(v)).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, wo, bo)


# ------------------------------------------------------------------------------------------
# third-party: UNet2DConditionModel (diffusers 0.14), SURVEY App. A.1  — PARITY UNPINNED
# ------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    e = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)  # flip_sin_to_cos=True


def _resnet(w: W, p: str, x, temb, groups, eps):
    h = F.silu(F.group_norm(x, groups, w[p + "norm1.weight"], w[p + "norm1.bias"], eps))
    h = F.conv2d(h, w[p + "conv1.weight"], w[p + "conv1.bias"], padding=1)
    if temb is not None:
        h = h + F.linear(F.silu(temb), w[p + "time_emb_proj.weight"], w[p + "time_emb_proj.bias"])[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, w[p + "norm2.weight"], w[p + "norm2.bias"], eps))
    h = F.conv2d(h, w[p + "conv2.weight"], w[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in w:
        x = F.conv2d(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"])
    return x + h


def _transformer(w: W, p: str, x, ctx, heads, groups, linear_proj):
    B, C, H, Wd = x.shape
    res = x
    h = F.group_norm(x, groups, w[p + "norm.weight"], w[p + "norm.bias"], 1e-6)
    if linear_proj:
        h = h.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
        h = F.linear(h, w[p + "proj_in.weight"], w[p + "proj_in.bias"])
    else:
        h = F.conv2d(h, w[p + "proj_in.weight"], w[p + "proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
    t = p + "transformer_blocks.0."
    n = F.layer_norm(h, (C,), w[t + "norm1.weight"], w[t + "norm1.bias"])
    h = h + xti_attention(w[t + "attn1.to_q.weight"], w[t + "attn1.to_k.weight"], w[t + "attn1.to_v.weight"],
                          w[t + "attn1.to_out.0.weight"], w[t + "attn1.to_out.0.bias"], heads, n, None)
    n = F.layer_norm(h, (C,), w[t + "norm2.weight"], w[t + "norm2.bias"])
    h = h + xti_attention(w[t + "attn2.to_q.weight"], w[t + "attn2.to_k.weight"], w[t + "attn2.to_v.weight"],
                          w[t + "attn2.to_out.0.weight"], w[t + "attn2.to_out.0.bias"], heads, n, ctx)
    n = F.layer_norm(h, (C,), w[t + "norm3.weight"], w[t + "norm3.bias"])
    g = F.linear(n, w[t + "ff.net.0.proj.weight"], w[t + "ff.net.0.proj.bias"])
    a, gate = g.chunk(2, dim=-1)
    h = h + F.linear(a * F.gelu(gate), w[t + "ff.net.2.weight"], w[t + "ff.net.2.bias"])
    if linear_proj:
        h = F.linear(h, w[p + "proj_out.weight"], w[p + "proj_out.bias"])
        h = h.reshape(B, H, Wd, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, Wd, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, w[p + "proj_out.weight"], w[p + "proj_out.bias"])
    return h + res


def unet_forward(w: W, cfg, sample: torch.Tensor, timesteps: torch.Tensor, ctx) -> torch.Tensor:
    """sample (B,4,h,w), timesteps (B,), ctx = XTI dict or tensor -> predicted noise (B,4,h,w)."""
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    boc = cfg.block_out_channels
    temb = timestep_embedding(timesteps, boc[0])
    temb = F.linear(temb, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])
    h = F.conv2d(sample, w["conv_in.weight"], w["conv_in.bias"], padding=1)
    skips = [h]
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            h = _resnet(w, f"down_blocks.{i}.resnets.{j}.", h, temb, G, eps)
            if cfg.down_has_attn[i]:
                h = _transformer(w, f"down_blocks.{i}.attentions.{j}.", h, ctx, cfg.num_heads[i], G,
                                 cfg.use_linear_projection)
            skips.append(h)
        if i < len(boc) - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv."
            h = F.conv2d(h, w[p + "weight"], w[p + "bias"], stride=2, padding=1)
            skips.append(h)
    h = _resnet(w, "mid_block.resnets.0.", h, temb, G, eps)
    h = _transformer(w, "mid_block.attentions.0.", h, ctx, cfg.num_heads[-1], G, cfg.use_linear_projection)
    h = _resnet(w, "mid_block.resnets.1.", h, temb, G, eps)
    up_has_attn = tuple(reversed(cfg.down_has_attn))
    up_heads = tuple(reversed(cfg.num_heads))
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet(w, f"up_blocks.{i}.resnets.{j}.", h, temb, G, eps)
            if up_has_attn[i]:
                h = _transformer(w, f"up_blocks.{i}.attentions.{j}.", h, ctx, up_heads[i], G,
                                 cfg.use_linear_projection)
        if i < len(boc) - 1:
            p = f"up_blocks.{i}.upsamplers.0.conv."
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, w[p + "weight"], w[p + "bias"], padding=1)
    h = F.silu(F.group_norm(h, G, w["conv_norm_out.weight"], w["conv_norm_out.bias"], eps))
    return F.conv2d(h, w["conv_out.weight"], w["conv_out.bias"], padding=1)


# ------------------------------------------------------------------------------------------
# third-party: AutoencoderKL.encode (diffusers 0.14), SURVEY App. A.3 — PARITY UNPINNED
# ------------------------------------------------------------------------------------------
def vae_encode_moments(w: W, cfg, x: torch.Tensor) -> torch.Tensor:
    """x (B,3,H,W) in [-1,1] -> moments (B, 2*latent, H/8, W/8) = quant_conv(encoder(x))."""
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    boc = cfg.block_out_channels
    h = F.conv2d(x, w["encoder.conv_in.weight"], w["encoder.conv_in.bias"], padding=1)
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            h = _resnet(w, f"encoder.down_blocks.{i}.resnets.{j}.", h, None, G, eps)
        if i < len(boc) - 1:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv."
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), w[p + "weight"], w[p + "bias"], stride=2)
    h = _resnet(w, "encoder.mid_block.resnets.0.", h, None, G, eps)
    a = "encoder.mid_block.attentions.0."
    B, C, H, Wd = h.shape
    res = h
    n = F.group_norm(h, G, w[a + "group_norm.weight"], w[a + "group_norm.bias"], eps)
    n = n.view(B, C, H * Wd).transpose(1, 2)
    q = F.linear(n, w[a + "query.weight"], w[a + "query.bias"])
    k = F.linear(n, w[a + "key.weight"], w[a + "key.bias"])
    v = F.linear(n, w[a + "value.weight"], w[a + "value.bias"])
    probs = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(C), -1)
    o = F.linear(probs @ v, w[a + "proj_attn.weight"], w[a + "proj_attn.bias"])
    h = o.transpose(1, 2).reshape(B, C, H, Wd) + res
    h = _resnet(w, "encoder.mid_block.resnets.1.", h, None, G, eps)
    h = F.silu(F.group_norm(h, G, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"], eps))
    h = F.conv2d(h, w["encoder.conv_out.weight"], w["encoder.conv_out.bias"], padding=1)
    return F.conv2d(h, w["quant_conv.weight"], w["quant_conv.bias"])


def gaussian_sample(moments: torch.Tensor, eps_noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample with an externally supplied N(0,1) draw."""
    mean, logvar = moments.chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * eps_noise


# ------------------------------------------------------------------------------------------
# third-party: DDPMScheduler, SURVEY App. A.4 — PARITY UNPINNED
# ------------------------------------------------------------------------------------------
def alphas_cumprod(cfg) -> torch.Tensor:
    betas = torch.linspace(cfg.beta_start ** 0.5, cfg.beta_end ** 0.5, cfg.num_train_timesteps,
                           dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(ac: torch.Tensor, x0, noise, t):
    a = ac[t] ** 0.5
    s = (1 - ac[t]) ** 0.5
    return a[:, None, None, None] * x0 + s[:, None, None, None] * noise


def get_velocity(ac: torch.Tensor, x0, noise, t):
    a = ac[t] ** 0.5
    s = (1 - ac[t]) ** 0.5
    return a[:, None, None, None] * noise - s[:, None, None, None] * x0


# ------------------------------------------------------------------------------------------
# the train step (Coach.train body, training/coach.py:154-218) for learnable_mode 0
# ------------------------------------------------------------------------------------------
def text_conditioning(clip_w: W, clip_cfg, mapper_p: W, w_enc, norm_scale, input_ids, placeholder_object,
                      timesteps, alpha=0.2, unconstrained=False, n_layers=16, view=None, hidden_masks=None,
                      output_bypass=True):
    """Coach.get_text_conditioning (training/coach.py:276-311): one text-encoder pass per UNet
    cross-attention layer; returns the XTI context dict.
    view = optional dict(p, w_enc, norm_scale, placeholder, params, alpha, unconstrained, hidden_masks).
    hidden_masks: optional [n_layers, B, hidden] 0/1 nested-dropout masks (neti_mapper.py:401-414)."""
    hs = {"this_idx": 0}
    B = input_ids.shape[0]
    for l in range(n_layers):
        layer = torch.full((B,), float(l))
        tm = None if hidden_masks is None else hidden_masks[l]
        if "input_layer.weight" in mapper_p:  # legacy object mapper (arch_view_net <= 14): w_enc is NeTIPositionalEncoding.w
            word, byp = mapper_forward_legacy(mapper_p, w_enc, timesteps, layer, norm_scale, output_bypass,
                                              truncation_mask=tm)
        else:
            word, byp = mapper_forward(mapper_p, w_enc, timesteps, layer, norm_scale, output_bypass, n_layers,
                                       truncation_mask=tm)
        kw = {}
        if view is not None:
            vm = view.get("hidden_masks")
            wv, bv = mapper_forward(view["p"], view["w_enc"], timesteps, layer, view["norm_scale"],
                                    view.get("output_bypass", True), n_layers, view_params=view["params"],
                                    truncation_mask=None if vm is None else vm[l])
            kw = dict(placeholder_view=view["placeholder"], word_view=wv, bypass_view=bv,
                      unconstrained_view=view.get("unconstrained", False), alpha_view=view.get("alpha", alpha))
        last, last_b = neti_text_encoder(clip_w, clip_cfg, input_ids, placeholder_object, word, byp,
                                         unconstrained, alpha, **kw)
        hs[f"CONTEXT_TENSOR_{l}"] = last
        if last_b is not None:  # no mapper with output_bypass: the key is absent (coach.py:300-304), V reads CONTEXT_TENSOR
            hs[f"CONTEXT_TENSOR_BYPASS_{l}"] = last_b
    return hs


def train_step_loss(sd_cfg, unet_w, vae_w, clip_w, mapper_p, w_enc, norm_scale, pixel_values, input_ids,
                    placeholder_object, timesteps, eps_latent, noise, alpha=0.2, unconstrained=False,
                    ctx_round=None, view=None):
    """Forward of one micro-step; returns (loss, aux).  `ctx_round` optionally rounds the context
    tensors (e.g. lambda t: t.half().float()) to mimic the reference's `.to(weight_dtype)`
    (training/coach.py:299-304)."""
    with torch.no_grad():
        moments = vae_encode_moments(vae_w, sd_cfg.vae, pixel_values)
        latents = gaussian_sample(moments, eps_latent) * sd_cfg.vae.scaling_factor
        ac = alphas_cumprod(sd_cfg.ddpm)
        noisy = add_noise(ac, latents, noise, timesteps)
    hs = text_conditioning(clip_w, sd_cfg.clip, mapper_p, w_enc, norm_scale, input_ids, placeholder_object,
                           timesteps, alpha, unconstrained, sd_cfg.unet.n_cross_layers, view=view)
    if ctx_round is not None:
        for k in list(hs):
            if k != "this_idx" and hs[k] is not None:
                hs[k] = ctx_round(hs[k])
    pred = unet_forward(unet_w, sd_cfg.unet, noisy, timesteps, hs)
    if sd_cfg.ddpm.prediction_type == "epsilon":
        target = noise
    else:
        target = get_velocity(ac, latents, noise, timesteps)
    loss = F.mse_loss(pred.float(), target.float(), reduction="mean")
    return loss, dict(latents=latents, noisy=noisy, pred=pred, ctx=hs, moments=moments)


def adamw_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2):
    """torch.optim.AdamW (decoupled weight decay) single-tensor update; training/coach.py:750-756."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return p - lr * mhat / (vhat.sqrt() + eps), m, v

# ------------------------------------------------------------------------------------------
# inference path: AutoencoderKL.decode, DPM-Solver++(2M) / DDIM, sd_pipeline_call
# third-party pieces follow diffusers 0.14 — PARITY UNPINNED (diffusers is not installable here)
# ------------------------------------------------------------------------------------------
def vae_decode(w: W, cfg, z: torch.Tensor) -> torch.Tensor:
    """z (B,latent,h,w) already divided by the scaling factor -> image (B,3,8h,8w) in ~[-1,1]
    (`pipeline.decode_latents`, sd_pipeline_call.py:115: post_quant_conv -> Decoder)."""
    G, eps = cfg.norm_num_groups, cfg.norm_eps
    boc = list(reversed(cfg.block_out_channels))
    h = F.conv2d(z, w["post_quant_conv.weight"], w["post_quant_conv.bias"])
    h = F.conv2d(h, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"], padding=1)
    h = _resnet(w, "decoder.mid_block.resnets.0.", h, None, G, eps)
    a = "decoder.mid_block.attentions.0."
    B, C, H, Wd = h.shape
    n = F.group_norm(h, G, w[a + "group_norm.weight"], w[a + "group_norm.bias"], eps)
    n = n.view(B, C, H * Wd).transpose(1, 2)
    q = F.linear(n, w[a + "query.weight"], w[a + "query.bias"])
    k = F.linear(n, w[a + "key.weight"], w[a + "key.bias"])
    v = F.linear(n, w[a + "value.weight"], w[a + "value.bias"])
    probs = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(C), -1)
    o = F.linear(probs @ v, w[a + "proj_attn.weight"], w[a + "proj_attn.bias"])
    h = o.transpose(1, 2).reshape(B, C, H, Wd) + h
    h = _resnet(w, "decoder.mid_block.resnets.1.", h, None, G, eps)
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(w, f"decoder.up_blocks.{i}.resnets.{j}.", h, None, G, eps)
        if i < len(boc) - 1:
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv."
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), w[p + "weight"], w[p + "bias"], padding=1)
    h = F.silu(F.group_norm(h, G, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(h, w["decoder.conv_out.weight"], w["decoder.conv_out.bias"], padding=1)


def inference_timesteps(kind: str, num_steps: int, num_train: int = 1000):
    """DPMSolverMultistepScheduler.set_timesteps / DDIMScheduler.set_timesteps (steps_offset=1 for SD)."""
    import numpy as np
    if kind == "dpm++2m":
        return [int(t) for t in np.linspace(0, num_train - 1, num_steps + 1).round()[::-1][:-1].astype(np.int64)]
    if kind == "ddim":
        ratio = num_train // num_steps
        return [int(t) for t in ((np.arange(0, num_steps) * ratio).round()[::-1].astype(np.int64) + 1)]
    raise ValueError(kind)


def step_coefficients(kind: str, ac: torch.Tensor, timesteps, i: int):
    """One sampler step written as x_prev = cx*x + c0*x0(t_i) + c1*x0(t_{i-1}) on the data prediction
    x0 (DPM-Solver++ `convert_model_output`; for DDIM with eta=0 the same algebra with c1 = 0).
    Returns (cx, c0, c1, alpha_t, sigma_t) as python floats; ac = alphas_cumprod (f32)."""
    ac = ac.double()
    t = timesteps[i]
    al, sg = ac.sqrt(), (1 - ac).sqrt()
    if kind == "ddim":
        ratio = 1000 // len(timesteps)
        tp = t - ratio
        a_t = ac[t]
        a_p = ac[tp] if tp >= 0 else ac[0]  # set_alpha_to_one=False in the SD scheduler configs
        r = ((1 - a_p) / (1 - a_t)).sqrt()
        return float(r), float(a_p.sqrt() - r * a_t.sqrt()), 0.0, float(al[t]), float(sg[t])
    lam = al.log() - sg.log()
    n = len(timesteps)
    tp = 0 if i == n - 1 else timesteps[i + 1]
    h = lam[tp] - lam[t]
    cx = sg[tp] / sg[t]
    base = -al[tp] * (torch.exp(-h) - 1.0)
    first_order = i == 0 or (i == n - 1 and n < 15)  # lower_order_nums < 1, lower_order_final
    if first_order:
        return float(cx), float(base), 0.0, float(al[t]), float(sg[t])
    h0 = lam[t] - lam[timesteps[i - 1]]
    r0 = h0 / h
    # D0 = m0, D1 = (m0 - m1)/r0:  x = cx*x + base*D0 + 0.5*base*D1
    return float(cx), float(base * (1 + 0.5 / r0)), float(-0.5 * base / r0), float(al[t]), float(sg[t])


def sampler_step(sd_cfg, kind, ac, ts, i, x, e, m_prev):
    """one `scheduler.step(noise_pred, t, latents).prev_sample` (DPM-Solver++(2M) / DDIM eta 0) on the data prediction;
    returns (x_prev, x0) — x0 is the history entry the next multistep update reads"""
    cx, c0, c1, a_t, s_t = step_coefficients(kind, ac, ts, i)
    x0 = (x - s_t * e) / a_t if sd_cfg.ddpm.prediction_type == "epsilon" else a_t * x - s_t * e
    return cx * x + c0 * x0 + c1 * m_prev, x0


def sd_pipeline_call(sd_cfg, unet_w, vae_dec_w, prompt_embeds, negative_embeds, latents, kind="dpm++2m",
                     num_inference_steps=50, guidance_scale=7.5, unet_fn=None, decode_fn=None):
    """sd_pipeline_call.py:8-133: per-step NeTI context dicts (`prompt_embeds[i]` when a list, the same object every
    step otherwise, :86), an unconditional embedding used for K and V of every layer FIRST (:75-81), then the conditional
    pass (:87-92), CFG (:96), sampler step (:99), decode, (x/2+0.5).clamp(0,1).
    prompt_embeds: list (one per timestep) of XTI context dicts, or one dict / tensor; negative_embeds (B,77,D).
    unet_fn(x, t, encoder_hidden_states) / decode_fn(latents): stand-ins for the two diffusers modules — the golden fixture
    G10 (oracle/make_golden.py) drives the REAL reference loop and this restatement with the same toy pair."""
    ac = alphas_cumprod(sd_cfg.ddpm)
    ts = inference_timesteps(kind, num_inference_steps, sd_cfg.ddpm.num_train_timesteps)
    if unet_fn is None:
        unet_fn = lambda x_, t_, hs_: unet_forward(unet_w, sd_cfg.unet, x_, torch.full((x_.shape[0],), int(t_),
                                                                                    dtype=torch.int64), hs_)
    x = latents.clone()
    m_prev = torch.zeros_like(x)
    for i, t in enumerate(ts):
        eu = unet_fn(x, t, negative_embeds)
        hs = prompt_embeds[i] if type(prompt_embeds) == list else prompt_embeds
        if isinstance(hs, dict) and "this_idx" not in hs:  # prompt_manager.py:77 starts every dict at 0; the 16
            hs = dict(hs, this_idx=0)                       # processors of one forward leave it at 0 again
        ec = unet_fn(x, t, hs)
        e = eu + guidance_scale * (ec - eu)
        x, m_prev = sampler_step(sd_cfg, kind, ac, ts, i, x, e, m_prev)
    if decode_fn is not None:
        return decode_fn(x), x
    img = vae_decode(vae_dec_w, sd_cfg.vae, x / sd_cfg.vae.scaling_factor)
    return (img / 2 + 0.5).clamp(0, 1), x
