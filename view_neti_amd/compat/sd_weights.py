"""Where the frozen SD weights come from (training/coach.py:600-640 loads them with from_pretrained).

`load_sd_weights(cfg, path, device)`:
  * `path` is a local diffusers-layout directory (unet/, vae/, text_encoder/ with *.safetensors):
    tensors are read by their diffusers/transformers state-dict names (sd_config.py enumerates them;
    the post-0.14 VAE attention names to_q/to_k/to_v/to_out.0 are mapped back to query/key/value/proj_attn);
  * otherwise (hub ids like "CompVis/stable-diffusion-v1-4" cannot be resolved offline) this is an ERROR, unless the
    caller opts in to SD-shaped synthetic weights from the counter-hash generator (`allow_synthetic=True`, i.e.
    `model.allow_synthetic_weights` in the config, or VNETI_ALLOW_SYNTHETIC_WEIGHTS=1 in the environment —
    benchmarks and tests): a run on random weights writes plausible-looking but meaningless checkpoints, so it
    must never happen by accident.
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


def synthetic_allowed(flag: bool = False) -> bool:
    return bool(flag) or os.environ.get("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "") not in ("", "0")


def _refuse(path: str, what: str):
    raise FileNotFoundError(
        f"'{path}' is not a local diffusers-layout checkpoint directory ({what} not found) and hub ids cannot be "
        "resolved offline. Point model.pretrained_model_name_or_path at a local directory, or opt in to SD-shaped "
        "SYNTHETIC weights with --model.allow_synthetic_weights true (or VNETI_ALLOW_SYNTHETIC_WEIGHTS=1).")


def load_sd_weights(cfg: sc.SDConfig, path: str, device: str = "cuda", allow_synthetic: bool = False
                    ) -> Tuple[Dict, Dict, Dict, bool]:
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
    if not synthetic_allowed(allow_synthetic):
        _refuse(path, "unet/")
    return (synth.unet_weights(cfg.unet, device=device), synth.vae_weights(cfg.vae, device=device),
            synth.clip_weights(cfg.clip, device=device), True)


def load_vae_decoder_weights(cfg: sc.SDConfig, path: str, device: str = "cuda", allow_synthetic: bool = False
                             ) -> Tuple[Dict, bool]:
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
    if not synthetic_allowed(allow_synthetic):
        _refuse(path, "vae/")
    return synth.vae_decoder_weights(cfg.vae, device=device), True
