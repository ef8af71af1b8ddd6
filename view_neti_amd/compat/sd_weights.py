"""Where the frozen SD weights come from (training/coach.py:600-640 loads them with from_pretrained).

`load_sd_weights(cfg, path, device)`:
  * `path` is a local diffusers-layout directory (unet/, vae/, text_encoder/ with *.safetensors):
    tensors are read by their diffusers/transformers state-dict names (sd_config.py enumerates them;
    the post-0.14 VAE attention names to_q/to_k/to_v/to_out.0 are mapped back to query/key/value/proj_attn);
  * otherwise (hub ids like "CompVis/stable-diffusion-v1-4" cannot be resolved offline) SD-shaped
    synthetic weights from the counter-hash generator are returned and the caller is told so.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

from .. import sd_config as sc
from .. import synth

_VAE_RENAME = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}


def _read_dir(d: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".safetensors"):
            out.update(load_file(os.path.join(d, f)))
    return out


def load_sd_weights(cfg: sc.SDConfig, path: str, device: str = "cuda") -> Tuple[Dict, Dict, Dict, bool]:
    """-> (unet, vae, clip, is_synthetic)"""
    if path and os.path.isdir(os.path.join(str(path), "unet")):
        unet = _read_dir(os.path.join(path, "unet"))
        vae_raw = _read_dir(os.path.join(path, "vae"))
        vae = {}
        for k, v in vae_raw.items():
            for new, old in _VAE_RENAME.items():
                k = k.replace(f"attentions.0.{new}.", f"attentions.0.{old}.")
            vae[k] = v
        clip = _read_dir(os.path.join(path, "text_encoder"))
        for name, need, have in (("unet", sc.unet_shapes(cfg.unet), unet), ("vae", sc.vae_encoder_shapes(cfg.vae), vae),
                                 ("text_encoder", sc.clip_text_shapes(cfg.clip), clip)):
            missing = [k for k in need if k not in have]
            if missing:
                raise KeyError(f"{name}: {len(missing)} tensors missing from checkpoint, e.g. {missing[:3]}")
        f32 = lambda d, need: {k: d[k].float() for k in need}
        return (f32(unet, sc.unet_shapes(cfg.unet)), f32(vae, sc.vae_encoder_shapes(cfg.vae)),
                f32(clip, sc.clip_text_shapes(cfg.clip)), False)
    return (synth.unet_weights(cfg.unet, device=device), synth.vae_weights(cfg.vae, device=device),
            synth.clip_weights(cfg.clip, device=device), True)


def load_vae_decoder_weights(cfg: sc.SDConfig, path: str, device: str = "cuda") -> Tuple[Dict, bool]:
    """`post_quant_conv.*` + `decoder.*` of the same checkpoint directory (inference path); synthetic otherwise."""
    need = sc.vae_decoder_shapes(cfg.vae)
    if path and os.path.isdir(os.path.join(str(path), "vae")):
        raw = _read_dir(os.path.join(path, "vae"))
        vae = {}
        for k, v in raw.items():
            for new, old in _VAE_RENAME.items():
                k = k.replace(f"attentions.0.{new}.", f"attentions.0.{old}.")
            vae[k] = v
        missing = [k for k in need if k not in vae]
        if missing:
            raise KeyError(f"vae decoder: {len(missing)} tensors missing from checkpoint, e.g. {missing[:3]}")
        return {k: vae[k].float().reshape(need[k]) for k in need}, False
    return synth.vae_decoder_weights(cfg.vae, device=device), True
