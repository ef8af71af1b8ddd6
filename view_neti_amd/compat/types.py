"""The three dataclasses of the reference's `utils/types.py:8-31`.  `PESigmas` lives in compat/config.py
(its defaults are None rather than the reference's `= float` typo); `NeTIBatch` is what `text_encoder(batch=...)`
receives (models/neti_clip_text_encoder.py:23-42), `MapperOutput` what `NeTIMapper.forward` returns
(models/neti_mapper.py:165-197, :416-438)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .config import PESigmas  # noqa: F401  (re-exported under the reference's module path)


@dataclass
class NeTIBatch:
    input_ids: torch.Tensor
    input_ids_placeholder_object: torch.Tensor
    input_ids_placeholder_view: torch.Tensor
    timesteps: torch.Tensor
    unet_layers: torch.Tensor
    truncation_idx: Optional[int] = None


@dataclass
class MapperOutput:
    word_embedding: torch.Tensor
    bypass_output: torch.Tensor
    bypass_unconstrained: bool
    output_bypass_alpha: float
