"""TEST INFRASTRUCTURE ONLY (imported by tests/; nothing under view_neti_amd/ may import it).

A second, independent statement of the diffusers-0.14 graphs the train step runs — `UNet2DConditionModel` with the XTI
attention processor, and the encoder half of `AutoencoderKL` — this time as a tree of `torch.nn` MODULES laid out class by
class and attribute by attribute like the library (`ResnetBlock2D.norm1/conv1/time_emb_proj/norm2/conv2/conv_shortcut`,
`Transformer2DModel.norm/proj_in/transformer_blocks[0]/proj_out`, `BasicTransformerBlock.norm1/attn1/norm2/attn2/norm3/ff`,
`CrossAttention.to_q/to_k/to_v/to_out[0]`, `FeedForward.net[0].proj / net[2]`, `Downsample2D.conv`, `Upsample2D.conv`,
`CrossAttnDownBlock2D.resnets/attentions/downsamplers`, `UNetMidBlock2DCrossAttn`, `AttentionBlock.group_norm/query/key/
value/proj_attn`, `Encoder.conv_in/down_blocks/mid_block/conv_norm_out/conv_out`, `quant_conv`).

What it pins (diffusers itself is absent from /root/reference and not installable here, so `oracle/sd_ref.py`'s
functional restatement of these graphs was pinned by nothing runnable):
  * the parameter NAMES and SHAPES come out of `nn.Module` registration here, not out of strings — a
    `load_state_dict(strict=True)` of the state dict `view_neti_amd.sd_config.unet_shapes / vae_encoder_shapes`
    describe (and `synth.py` fills) fails on any missing, extra or mis-shaped key, at the tiny AND the published sizes;
  * the arithmetic goes through torch's own layers (`nn.GroupNorm`, `nn.Conv2d`, `nn.LayerNorm`, `nn.Linear`,
    `F.scaled_dot_product_attention`) instead of the functional calls and the hand-written softmax(QK^T)V of sd_ref.py.
Reference call sites: training/coach.py:165-169 (vae.encode), :197-198 (unet), models/xti_attention_processor.py:9-57.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        if temb is not None:
            self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if temb is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x) + h


class CrossAttention(nn.Module):
    """diffusers CrossAttention as XTIAttenProc drives it: K and V may come from different tensors."""

    def __init__(self, query_dim, context_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(query_dim, query_dim, bias=False)
        self.to_k = nn.Linear(context_dim, query_dim, bias=False)
        self.to_v = nn.Linear(context_dim, query_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim), nn.Dropout(0.0)])

    def forward(self, hidden, ctx_k=None, ctx_v=None):
        B, N, C = hidden.shape
        ctx_k = hidden if ctx_k is None else ctx_k
        ctx_v = ctx_k if ctx_v is None else ctx_v
        split = lambda t: t.view(B, t.shape[1], self.heads, C // self.heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.to_q(hidden)), split(self.to_k(ctx_k)), split(self.to_v(ctx_v)))
        return self.to_out[1](self.to_out[0](o.transpose(1, 2).reshape(B, N, C)))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        a, gate = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, context_dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = CrossAttention(dim, dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, h, ctx_k, ctx_v):
        h = h + self.attn1(self.norm1(h))
        h = h + self.attn2(self.norm2(h), ctx_k, ctx_v)
        return h + self.ff(self.norm3(h))


class Transformer2DModel(nn.Module):
    def __init__(self, dim, context_dim, heads, groups, linear_proj):
        super().__init__()
        self.linear_proj = linear_proj
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, context_dim, heads)])
        self.proj_out = nn.Linear(dim, dim) if linear_proj else nn.Conv2d(dim, dim, 1)

    def forward(self, x, ctx_k, ctx_v):
        B, C, H, W = x.shape
        h = self.norm(x)
        if self.linear_proj:
            h = self.proj_in(h.permute(0, 2, 3, 1).reshape(B, H * W, C))
        else:
            h = self.proj_in(h).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.transformer_blocks[0](h, ctx_k, ctx_v)
        if self.linear_proj:
            h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.reshape(B, H, W, C).permute(0, 3, 1, 2))
        return h + x


class _Sampler(nn.Module):  # Downsample2D / Upsample2D: one attribute, `conv`
    def __init__(self, c, stride):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=1)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()
        self.resnets = nn.ModuleList()


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, temb):
        super().__init__()
        self.linear_1 = nn.Linear(cin, temb)
        self.linear_2 = nn.Linear(temb, temb)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class UNet2DConditionModel(nn.Module):
    """cfg: view_neti_amd.sd_config.UNetConfig.  forward(sample, timesteps, ctx) with ctx the XTI dict
    {"this_idx", "CONTEXT_TENSOR_i", "CONTEXT_TENSOR_BYPASS_i"} (prompt_manager.py:79-99) or one tensor."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        boc, G, eps, temb = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps, cfg.temb_dim
        Dc, lin = cfg.cross_attention_dim, cfg.use_linear_projection
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        cin = boc[0]
        for i, cout in enumerate(boc):
            blk = _Block()
            if cfg.down_has_attn[i]:
                blk.attentions = nn.ModuleList()
            for j in range(cfg.layers_per_block):
                blk.resnets.append(ResnetBlock2D(cin if j == 0 else cout, cout, temb, G, eps))
                if cfg.down_has_attn[i]:
                    blk.attentions.append(Transformer2DModel(cout, Dc, cfg.num_heads[i], G, lin))
            if i < len(boc) - 1:
                blk.downsamplers = nn.ModuleList([_Sampler(cout, 2)])
            self.down_blocks.append(blk)
            cin = cout
        cm = boc[-1]
        self.mid_block = _Block()
        self.mid_block.attentions = nn.ModuleList([Transformer2DModel(cm, Dc, cfg.num_heads[-1], G, lin)])
        self.mid_block.resnets.extend([ResnetBlock2D(cm, cm, temb, G, eps), ResnetBlock2D(cm, cm, temb, G, eps)])
        self.up_blocks = nn.ModuleList()
        rev, rev_attn, rev_heads = tuple(reversed(boc)), tuple(reversed(cfg.down_has_attn)), tuple(reversed(cfg.num_heads))
        n = cfg.layers_per_block + 1
        for i, out in enumerate(rev):
            blk = _Block()
            if rev_attn[i]:
                blk.attentions = nn.ModuleList()
            prev = rev[i - 1] if i > 0 else rev[0]
            inp = rev[min(i + 1, len(rev) - 1)]
            for j in range(n):  # diffusers get_up_block: the last resnet of a block takes the skip of the level below
                skip = inp if j == n - 1 else out
                rin = prev if j == 0 else out
                blk.resnets.append(ResnetBlock2D(rin + skip, out, temb, G, eps))
                if rev_attn[i]:
                    blk.attentions.append(Transformer2DModel(out, Dc, rev_heads[i], G, lin))
            if i < len(rev) - 1:
                blk.upsamplers = nn.ModuleList([_Sampler(out, 1)])
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(G, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    @staticmethod
    def _ctx(ctx):
        """XTIAttenProc (models/xti_attention_processor.py:16-42,53-55): K from CONTEXT_TENSOR_i, V from the BYPASS tensor
        when present, the index advancing with every cross-attention call and wrapping after the last one."""
        if not isinstance(ctx, dict):
            return ctx, ctx
        i = ctx["this_idx"]
        k = ctx[f"CONTEXT_TENSOR_{i}"]
        v = ctx.get(f"CONTEXT_TENSOR_BYPASS_{i}", k)
        ctx["this_idx"] = i + 1
        return k, v

    def forward(self, sample, timesteps, ctx):
        cfg = self.cfg
        if isinstance(ctx, dict):
            ctx = dict(ctx)
            ctx["this_idx"] = 0
        half = cfg.block_out_channels[0] // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        e = timesteps.float()[:, None] * freqs[None]
        temb = self.time_embedding(torch.cat([torch.cos(e), torch.sin(e)], dim=-1))  # flip_sin_to_cos, freq_shift 0
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            for j, res in enumerate(blk.resnets):
                h = res(h, temb)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j](h, *self._ctx(ctx))
                skips.append(h)
            if hasattr(blk, "downsamplers"):
                h = blk.downsamplers[0].conv(h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, temb)
        h = self.mid_block.attentions[0](h, *self._ctx(ctx))
        h = self.mid_block.resnets[1](h, temb)
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                h = res(torch.cat([h, skips.pop()], dim=1), temb)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j](h, *self._ctx(ctx))
            if hasattr(blk, "upsamplers"):
                h = blk.upsamplers[0].conv(F.interpolate(h, scale_factor=2.0, mode="nearest"))
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class AttentionBlock(nn.Module):
    """diffusers 0.14 AttentionBlock of the VAE mid block: one head of dim C."""

    def __init__(self, c, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.query, self.key, self.value, self.proj_attn = (nn.Linear(c, c) for _ in range(4))

    def forward(self, x):
        B, C, H, W = x.shape
        n = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        o = F.scaled_dot_product_attention(self.query(n)[:, None], self.key(n)[:, None], self.value(n)[:, None])[:, 0]
        return self.proj_attn(o).transpose(1, 2).reshape(B, C, H, W) + x


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, G, eps = cfg.block_out_channels, cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cin = boc[0]
        for i, cout in enumerate(boc):
            blk = _Block()
            for j in range(cfg.layers_per_block):
                blk.resnets.append(ResnetBlock2D(cin if j == 0 else cout, cout, None, G, eps))
            if i < len(boc) - 1:
                blk.downsamplers = nn.ModuleList([_Sampler(cout, 2)])
                blk.downsamplers[0].conv.padding = (0, 0)  # Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) first
            self.down_blocks.append(blk)
            cin = cout
        cm = boc[-1]
        self.mid_block = _Block()
        self.mid_block.attentions = nn.ModuleList([AttentionBlock(cm, G, eps)])
        self.mid_block.resnets.extend([ResnetBlock2D(cm, cm, None, G, eps), ResnetBlock2D(cm, cm, None, G, eps)])
        self.conv_norm_out = nn.GroupNorm(G, cm, eps=eps)
        self.conv_out = nn.Conv2d(cm, 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for blk in self.down_blocks:
            for res in blk.resnets:
                h = res(h)
            if hasattr(blk, "downsamplers"):
                h = blk.downsamplers[0].conv(F.pad(h, (0, 1, 0, 1)))
        h = self.mid_block.resnets[0](h)
        h = self.mid_block.attentions[0](h)
        h = self.mid_block.resnets[1](h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class AutoencoderKLEncoder(nn.Module):
    """`AutoencoderKL.encode(x)` up to the moments: quant_conv(encoder(x)).  cfg: sd_config.VAEConfig."""

    def __init__(self, cfg):
        super().__init__()
        self.encoder = _Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def forward(self, x):
        return self.quant_conv(self.encoder(x))


def state_shapes(module: nn.Module) -> Dict[str, tuple]:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}
