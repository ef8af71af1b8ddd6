"""`PromptManager` with the surface of the reference's prompt_manager.py:13-101.

The reference's `embed_prompt` runs the text encoder T x 16 times up front and returns T dicts of 32 tensors.
On the HIP engine the same conditioning is produced inside the denoising loop (16 layers per launch schedule,
one timestep at a time), so `embed_prompt` returns a light `PromptEmbeds` record — token ids, which placeholder
tokens the prompt holds, the camera parameters of its view token — that `sd_pipeline_call` hands to the engine.
Same argument names and meaning; the one-placeholder-per-kind check of prompt_manager.py:60-66 is kept.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

from .constants import UNET_LAYERS


@dataclass
class PromptEmbeds:
    text: str
    input_ids: torch.Tensor                     # (1, 77) int64
    input_ids_placeholder_object: torch.Tensor  # (1,) token id or -1
    input_ids_placeholder_view: torch.Tensor    # (1,) token id or -1
    view_params: Optional[torch.Tensor]         # (1, 12) scaled to [-1, 1], or None
    truncation_idx: Optional[int]
    num_images_per_prompt: int


class PromptManager:
    def __init__(self, tokenizer, text_encoder=None, timesteps: Optional[List[int]] = None,
                 unet_layers: List[str] = UNET_LAYERS, placeholder_view_token_ids: List[int] = None,
                 placeholder_object_token_ids: List[int] = None, torch_dtype: torch.dtype = torch.float16,
                 view_params_fn: Optional[Callable[[int], torch.Tensor]] = None):
        """text_encoder / timesteps / torch_dtype are accepted for signature compatibility (the engine owns the
        encoder and the sampler's timesteps).  view_params_fn(token_id) -> 12 scaled camera parameters."""
        self.tokenizer = tokenizer
        self.text_encoder = text_encoder
        self.timesteps = timesteps
        self.unet_layers = unet_layers
        self.placeholder_view_token_ids = list(placeholder_view_token_ids or [])
        self.placeholder_object_token_ids = list(placeholder_object_token_ids or [])
        self.dtype = torch_dtype
        self.view_params_fn = view_params_fn

    def embed_prompt(self, text: str, truncation_idx: Optional[int] = None,
                     num_images_per_prompt: int = 1) -> PromptEmbeds:
        ids = self.tokenizer(text, padding="max_length", max_length=self.tokenizer.model_max_length,
                             return_tensors="pt").input_ids

        def placeholder(token_ids):
            if not token_ids:
                return torch.tensor([-1])
            locs = torch.isin(ids, torch.tensor(token_ids))
            if locs.sum() == 0:
                return torch.tensor([-1])
            assert locs.sum() == 1, f"should be exactly 1 placeholder token of a kind per prompt, for prompt [`{text}`]"
            return ids[torch.where(locs)]

        obj = placeholder(self.placeholder_object_token_ids)
        view = placeholder(self.placeholder_view_token_ids)
        vp = None
        if int(view) != -1:
            if self.view_params_fn is None:
                raise ValueError("a prompt with a view token needs view_params_fn (token id -> camera parameters)")
            vp = self.view_params_fn(int(view)).reshape(1, -1).float()
        return PromptEmbeds(text, ids, obj, view, vp, truncation_idx, num_images_per_prompt)
