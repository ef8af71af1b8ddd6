"""`sd_pipeline_call` with the argument surface of the reference's sd_pipeline_call.py:8-133, running on
`view_neti_amd.engine.infer.InferenceEngine` instead of a diffusers `StableDiffusionPipeline`.

    pipeline        -> an `InferencePipeline` (engine + tokenizer; built once per resolution/batch)
    prompt_embeds   -> EITHER the reference's own contract (sd_pipeline_call.py:86-92): a list of T per-step XTI dicts
                       (what the reference's PromptManager.embed_prompt returns, prompt_manager.py:79-99), one dict, or one
                       (B, 77, D) tensor — fed straight to the UNet's per-layer K / V sources, the engine's text pass skipped;
                       OR the light `PromptEmbeds` record of this package's PromptManager (conditioning computed inside
                       the loop, 16 layers per launch schedule)
    height / width  -> fixed at engine construction; passing different values raises
    scheduler       -> `pipeline.sampler` ("dpm++2m" as installed by validate.py:568, or "ddim")
Returns an object with `.images` (list of PIL images, `output_type="pil"`) or the array, like the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from ..engine.infer import InferenceEngine
from .prompt_manager import PromptEmbeds


@dataclass
class InferencePipeline:
    engine: InferenceEngine
    tokenizer: Any
    sampler: str = "dpm++2m"


@dataclass
class PipelineOutput:
    images: Any
    nsfw_content_detected: Optional[bool] = False


def get_neg_prompt_input_ids(pipeline: InferencePipeline, negative_prompt: Optional[Union[str, List[str]]] = None):
    """sd_pipeline_call.py:136-150: tokenizer(negative_prompt or "", padding="max_length", truncation=True)."""
    if negative_prompt is None:
        negative_prompt = ""
    toks = [negative_prompt] if isinstance(negative_prompt, str) else negative_prompt
    return pipeline.tokenizer(toks, padding="max_length", max_length=pipeline.tokenizer.model_max_length,
                              truncation=True, return_tensors="pt")


@torch.no_grad()
def sd_pipeline_call(pipeline: InferencePipeline, prompt_embeds: Union[PromptEmbeds, List[Dict[str, Any]], Dict[str, Any],
                                                                       torch.Tensor], height: Optional[int] = None,
                     width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                     negative_prompt: Optional[Union[str, List[str]]] = None, num_images_per_prompt: Optional[int] = 1,
                     eta: float = 0.0, generator: Optional[torch.Generator] = None,
                     latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "pil",
                     return_dict: bool = True):
    eng = pipeline.engine
    B = eng.B
    H, W = eng.h * 8, eng.w * 8
    if (height or H) != H or (width or W) != W:
        raise ValueError(f"the engine was built for {H}x{W}")
    if num_images_per_prompt != B:
        raise ValueError(f"the engine was built for {B} images per call (num_images_per_prompt={num_images_per_prompt})")
    if eta != 0.0:
        raise NotImplementedError("eta > 0 (stochastic DDIM) is not implemented")
    neg = get_neg_prompt_input_ids(pipeline, negative_prompt)
    eng.set_negative_prompt(neg.input_ids)
    if latents is None:  # pipeline.prepare_latents: randn(shape, generator) * init_noise_sigma (= 1)
        latents = torch.randn((B, eng.Lc, eng.h, eng.w), generator=generator, dtype=torch.float32)
    if isinstance(prompt_embeds, PromptEmbeds):
        rep = lambda t: None if t is None else t.expand(B, *t.shape[1:]) if t.dim() > 1 else t.expand(B)
        eng.set_prompt(rep(prompt_embeds.input_ids), rep(prompt_embeds.input_ids_placeholder_object),
                       rep(prompt_embeds.input_ids_placeholder_view), rep(prompt_embeds.view_params),
                       prompt_embeds.truncation_idx)
        out = eng.generate(latents.to(eng.dev), num_inference_steps, guidance_scale, pipeline.sampler,
                           decode=output_type != "latent")
    elif isinstance(prompt_embeds, (list, dict, torch.Tensor)):
        # the reference's contract: conditioning computed by the caller, `prompt_embeds[i]` per step when a list (:86)
        out = eng.generate_from_contexts(latents.to(eng.dev), prompt_embeds, num_inference_steps, guidance_scale,
                                         pipeline.sampler, decode=output_type != "latent")
    else:
        raise TypeError(f"prompt_embeds of type {type(prompt_embeds).__name__}: expected PromptEmbeds, a list of per-step "
                        "context dicts, one dict or one tensor")
    if output_type == "latent":
        image, nsfw = out.clone(), None
    else:
        image = out.cpu().numpy()  # (B, H, W, 3) f32 in [0,1] = decode_latents' output
        nsfw = False
        if output_type == "pil":
            from PIL import Image
            image = [Image.fromarray(im) for im in (image * 255).round().astype(np.uint8)]
    if not return_dict:
        return image, nsfw
    return PipelineOutput(images=image, nsfw_content_detected=nsfw)
