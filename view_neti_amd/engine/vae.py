"""AutoencoderKL encoder (+ quant_conv) as a static HIP launch schedule, forward only.

Reference call site: `self.vae.encode(latent_batch).latent_dist` (training/coach.py:165-168);
the VAE is frozen and its output is `.detach()`ed, so there is no backward.  Output: the
distribution moments [B*h*w, 2*latent] (mean | logvar), channels-last f16; sampling, scaling
and add-noise are fused in one later kernel (ops.sample_add_noise).

Notes
  * conv_in (3 channels) is its own kernel, straight from the NCHW f32 pixels (csrc/conv_in.hip; the 27->64 padded
    im2col + GEMM remains for other channel counts).
  * Downsample2D in the VAE pads (0,1,0,1) and uses padding=0: the implicit-GEMM loader's bounds
    check provides the bottom/right zeros.
  * the single-head d=512 mid-block attention uses batched MFMA GEMMs + a row-softmax kernel
    (scores are 4 x 4096 x 4096 f16 = 134 MB at 512^2; no backward is needed).
  * conv_out (512 -> 8) and quant_conv (1x1, 8 -> 8) are folded into one 3x3 conv at pack time.
"""
from __future__ import annotations

from functools import partial
from typing import Dict

import torch

from .. import ops, packing
from .. import sd_config as sc
from .schedule import Schedule, T, rup


class VAEEncoderEngine(Schedule):
    def __init__(self, cfg: sc.VAEConfig, weights: Dict[str, torch.Tensor], batch: int, height: int, width: int,
                 device: str = "cuda", autotune: bool = True):
        super().__init__(batch, cfg.norm_num_groups, cfg.norm_eps, device, need_backward=False)
        self.cfg = cfg
        self.H, self.W = height, width
        self.x_in = self._buf((batch, cfg.in_channels, height, width), torch.float32)
        nlev = len(cfg.block_out_channels)
        self.h_out, self.w_out = height >> (nlev - 1), width >> (nlev - 1)
        self.moments = self._buf((batch * self.h_out * self.w_out, 2 * cfg.latent_channels))
        self._build(weights)
        if autotune:
            self.autotune()
        self.bind_workspace()
        self.fuse_gn_stats()

    def _build(self, w):
        cfg = self.cfg
        B, H, W = self.B, self.H, self.W
        boc = cfg.block_out_channels
        M0 = B * H * W
        b_in = self._w32(w["encoder.conv_in.bias"])
        h0 = self._buf((M0, boc[0]))
        if cfg.in_channels <= 3 and boc[0] % 128 == 0:
            # straight from the pixels: no [pixels][64] im2col matrix for a layer that is pure output bandwidth
            w_in = self._w16(packing.conv_in_direct(w["encoder.conv_in.weight"]))
            self.fwd.append(partial(ops.conv3x3_in, self.x_in, w_in, b_in, h0, B, cfg.in_channels, H, W,
                                    self.x_in.stride()))
        else:
            col = self._buf((M0, 64))
            w_in = self._w16(packing.pad_rows(packing.conv3x3_fwd(w["encoder.conv_in.weight"]), 64))
            self.fwd.append(partial(ops.im2col3x3_small, self.x_in, col, B, cfg.in_channels, H, W, H, W, 1, 1, 1,
                                    self.x_in.stride()))
            self.fwd.append(partial(ops.gemm, col, w_in, h0, bias=b_in))
        hcur = self._produced(T(h0, need_grad=False))
        cin = boc[0]
        h, wd = H, W
        for i, cout in enumerate(boc):
            for j in range(cfg.layers_per_block):
                hcur = self._resnet(hcur, cin if j == 0 else cout, cout, f"encoder.down_blocks.{i}.resnets.{j}.", w,
                                    None, h, wd, need_dx=False)
            if i < len(boc) - 1:
                hcur = self._downsample(hcur, cout, f"encoder.down_blocks.{i}.downsamplers.0.conv.", w, None, h, wd,
                                        pad=0)
                h, wd = h // 2, wd // 2
            cin = cout
        cm = boc[-1]
        hcur = self._resnet(hcur, cm, cm, "encoder.mid_block.resnets.0.", w, None, h, wd, need_dx=False)
        hcur = self._mid_attention(hcur, cm, "encoder.mid_block.attentions.0.", w, h * wd)
        hcur = self._resnet(hcur, cm, cm, "encoder.mid_block.resnets.1.", w, None, h, wd, need_dx=False)
        n, _ = self._gn(hcur, "encoder.conv_norm_out", w, cfg.norm_eps, True)
        # fold quant_conv (1x1) into conv_out:  W' = Wq . Wco,  b' = Wq . bco + bq   (exact in real arithmetic)
        wq = w["quant_conv.weight"].reshape(2 * cfg.latent_channels, 2 * cfg.latent_channels).double()
        wco = w["encoder.conv_out.weight"].double()
        wf = torch.einsum("om,mcyx->ocyx", wq, wco).float()
        bf = (wq @ w["encoder.conv_out.bias"].double() + w["quant_conv.bias"].double()).float()
        w_o = self._w16(packing.conv3x3_fwd(wf))
        b_o = self._w32(bf)
        self.fwd.append(partial(ops.gemm, n, w_o, self.moments, bias=b_o, M=B * h * wd,
                                conv=self._conv_desc(h, wd, cm, h, wd, 1, 1, 0, cm)))

    def _mid_attention(self, x: T, Cc, name, w, N):
        """diffusers 0.14 AttentionBlock: GN -> q,k,v (Linear with bias) -> softmax(q k^T / sqrt(C)) v
        -> proj_attn -> + residual; single head of dim C."""
        B = self.B
        M = x.rows
        g, _ = self._gn(x, name + "group_norm", w, self.eps, False)
        wqkv = self._w16(torch.cat([w[name + "query.weight"], w[name + "key.weight"], w[name + "value.weight"]], 0))
        bqkv = self._w32(torch.cat([w[name + "query.bias"], w[name + "key.bias"], w[name + "value.bias"]], 0))
        qkv = self._buf((M, 3 * Cc))
        self.fwd.append(partial(ops.gemm, g, wqkv, qkv, bias=bqkv))
        q, k, v = qkv[:, :Cc], qkv[:, Cc:2 * Cc], qkv[:, 2 * Cc:]
        ldn = rup(N, 64)  # K of the P.V GEMM must be a multiple of 64
        scores = self._buf((B, N, ldn), zero=True)
        self.fwd.append(partial(ops.gemm, q, k, scores, alpha=Cc ** -0.5, batch=B, strideA=N * 3 * Cc,
                                strideB=N * 3 * Cc, strideC=N * ldn, M=N, N=N, K=Cc, lda=3 * Cc, ldc=ldn))
        self.fwd.append(partial(ops.softmax_rows, scores.view(B * N, ldn), B * N, N))
        vt = self._buf((B, Cc, ldn), zero=True)
        self.fwd.append(partial(ops.transpose, v, vt, N, Cc, B, 3 * Cc, N * 3 * Cc, ldn, Cc * ldn))
        o = self._buf((M, Cc))
        self.fwd.append(partial(ops.gemm, scores, vt, o, batch=B, strideA=N * ldn, strideB=Cc * ldn, strideC=N * Cc,
                                M=N, N=Cc, K=ldn, lda=ldn, ldc=Cc))
        wo = self._w16(w[name + "proj_attn.weight"])
        bo = self._w32(w[name + "proj_attn.bias"])
        out = T(self._buf((M, Cc)), need_grad=False)
        self.fwd.append(partial(ops.gemm, o, wo, out.v, bias=bo, resid=x.v))
        return self._produced(out)


class VAEDecoderEngine(VAEEncoderEngine):
    """AutoencoderKL.decode as a forward-only launch schedule (inference path, SURVEY §8 f1;
    `pipeline.decode_latents`, sd_pipeline_call.py:115): latents/scaling -> post_quant_conv -> conv_in -> mid
    (resnet, single-head attention, resnet) -> 4 up blocks of 3 resnets (+ nearest-2x upsample fused into the
    following 3x3 conv's loader) -> GN+SiLU -> conv_out -> (x/2+0.5).clamp(0,1).
    Input: `z_in` f32 NCHW [B,latent,h,w] (the sampler's latents); output: `image` f32 [B, 8h, 8w, 3] in [0,1]."""

    def __init__(self, cfg: sc.VAEConfig, weights: Dict[str, torch.Tensor], batch: int, h: int, w: int,
                 device: str = "cuda", autotune: bool = True):
        Schedule.__init__(self, batch, cfg.norm_num_groups, cfg.norm_eps, device, need_backward=False)
        self.cfg = cfg
        self.h, self.w = h, w
        nlev = len(cfg.block_out_channels)
        self.H, self.W = h << (nlev - 1), w << (nlev - 1)
        lc = cfg.latent_channels
        self.z_in = self._buf((batch, lc, h, w), torch.float32)
        self.image = self._buf((batch, self.H, self.W, cfg.in_channels), torch.float32)
        self._build_decoder(weights)
        if autotune:
            self.autotune()
        self.bind_workspace()
        self.fuse_gn_stats()

    def _build_decoder(self, w):
        cfg = self.cfg
        B, h, wd = self.B, self.h, self.w
        lc = cfg.latent_channels
        boc = list(reversed(cfg.block_out_channels))
        f = self.fwd
        zq = self._buf((B, lc, h, wd), torch.float32)
        wpq = self._w32(w["post_quant_conv.weight"].reshape(lc, lc))
        bpq = self._w32(w["post_quant_conv.bias"])
        f.append(partial(ops.conv1x1_nchw, self.z_in, wpq, bpq, zq, B, lc, lc, h * wd, 1.0 / cfg.scaling_factor))
        M = B * h * wd
        col = self._buf((M, 64))
        cm = boc[0]
        w_in = self._w16(packing.pad_rows(packing.conv3x3_fwd(w["decoder.conv_in.weight"]), 64))
        b_in = self._w32(w["decoder.conv_in.bias"])
        h0 = self._buf((M, cm))
        f.append(partial(ops.im2col3x3_small, zq, col, B, lc, h, wd, h, wd, 1, 1, 1, zq.stride()))
        f.append(partial(ops.gemm, col, w_in, h0, bias=b_in))
        cur = self._produced(T(h0, need_grad=False))
        cur = self._resnet(cur, cm, cm, "decoder.mid_block.resnets.0.", w, None, h, wd, need_dx=False)
        cur = self._mid_attention(cur, cm, "decoder.mid_block.attentions.0.", w, h * wd)
        cur = self._resnet(cur, cm, cm, "decoder.mid_block.resnets.1.", w, None, h, wd, need_dx=False)
        cin = cm
        for i, cout in enumerate(boc):
            for j in range(cfg.layers_per_block + 1):
                cur = self._resnet(cur, cin if j == 0 else cout, cout, f"decoder.up_blocks.{i}.resnets.{j}.", w, None,
                                   h, wd, need_dx=False)
            if i < len(boc) - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv."
                wu = self._w16(packing.conv3x3_fwd(w[p + "weight"]))
                bu = self._w32(w[p + "bias"])
                up = T(self._buf((4 * cur.rows, cout)), need_grad=False)
                f.append(partial(ops.gemm, cur.v, wu, up.v, bias=bu, M=4 * cur.rows,
                                 conv=self._conv_desc(h, wd, cout, 2 * h, 2 * wd, 1, 1, 1, cur.v.stride(0))))
                cur = self._produced(up)
                h, wd = 2 * h, 2 * wd
            cin = cout
        n, _ = self._gn(cur, "decoder.conv_norm_out", w, cfg.norm_eps, True)
        co = cfg.in_channels
        w_o = self._w16(packing.conv3x3_fwd(w["decoder.conv_out.weight"]))
        b_o = self._w32(w["decoder.conv_out.bias"])
        self.rgb = self._buf((B * h * wd, 8))  # 3 channels used, row stride 8
        f.append(partial(ops.gemm, n, w_o, self.rgb[:, :co], bias=b_o, M=B * h * wd,
                         conv=self._conv_desc(h, wd, boc[-1], h, wd, 1, 1, 0, boc[-1])))
        f.append(partial(ops.image_postprocess, self.rgb[:, :co], self.image, B * h * wd, co))
