"""Generation from a finished training run — the role of the reference's `training/inference_dtu.py:283-398`
(`load_stable_diffusion_model`, mapper loading through `CheckpointHandler.load_mapper`, `PromptManager`,
`sd_pipeline_call`) without its DTU evaluation harness (SURVEY §8 f4 stays out of scope).

    pipe, pm = load_inference(exp_dir, "mapper-final", batch=1)       # config.yaml + mapper-final_{object,view}.pt
    out = sd_pipeline_call(pipe, pm.embed_prompt("<view_…>. A photo of a <object>"), num_inference_steps=30,
                           generator=torch.Generator().manual_seed(0))
    out.images[0].save("img.png")

The tokenizer is rebuilt exactly like `Coach` does it (view tokens first, then object tokens: the token ids the
mappers were trained with), the frozen SD weights come from `model.pretrained_model_name_or_path` (a local
diffusers directory) or the synthetic generator, and the new embedding rows are initialised from the
super-category rows — they are never read, the mappers overwrite them, but the table must have the rows.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import torch
import yaml

from .. import sd_config as sc
from ..engine.infer import InferenceEngine
from . import config as cfgmod
from .checkpoint_handler import CheckpointHandler
from .coach import _sd_family
from .dataset import TextualInversionDataset
from .prompt_manager import PromptManager
from .sd_pipeline_call import InferencePipeline
from .sd_weights import load_sd_weights, load_vae_decoder_weights
from .tokenizer import load_tokenizer


def load_inference(exp_dir, mapper_stem: str = "mapper-final", batch: int = 1, height: Optional[int] = None,
                   width: Optional[int] = None, object_token: Optional[str] = None, sampler: str = "dpm++2m",
                   device: str = "cuda") -> Tuple[InferencePipeline, PromptManager]:
    exp_dir = Path(exp_dir)
    with (exp_dir / "config.yaml").open() as f:
        cfg = cfgmod.decode(cfgmod.RunConfig, CheckpointHandler.clean_config_dict(yaml.safe_load(f)))
    sd = _sd_family(cfg)
    tok = load_tokenizer(str(cfg.model.pretrained_model_name_or_path), sd.clip.vocab_size)
    obj_path, view_path = exp_dir / f"{mapper_stem}_object.pt", exp_dir / f"{mapper_stem}_view.pt"
    raw = torch.load(obj_path, map_location="cpu", weights_only=False)
    object_tokens = [e["placeholder_object_token"] for e in raw["mappers"].values()]
    view_tokens = []
    cam_fn = None
    mapper_view = None
    if view_path.exists():
        lut_tok, lut_par = TextualInversionDataset.dtu_generate_dset_cam_tokens_params()
        view_tokens = [lut_tok[k] for k in sorted(lut_tok)]
    tok.add_tokens(view_tokens + object_tokens)
    view_ids = tok.convert_tokens_to_ids(view_tokens) if view_tokens else []
    object_ids = tok.convert_tokens_to_ids(object_tokens)
    _, lookup = CheckpointHandler.load_mapper(obj_path, "object", object_tokens, object_ids)
    if view_path.exists():
        cams = torch.stack([lut_par[k] for k in sorted(lut_par)])
        mins, maxs = cams.min(0).values.flatten(), cams.max(0).values.flatten()
        _, mapper_view = CheckpointHandler.load_mapper(view_path, "view", cam_mins=mins, cam_maxs=maxs)
        id2tok = dict(zip(view_ids, view_tokens))

        def cam_fn(token_id: int) -> torch.Tensor:
            p = TextualInversionDataset.dtu_token_to_cam_params(id2tok[token_id])[0]
            return (p - mins) / (maxs - mins) * 2 - 1
    object_token = object_token or object_tokens[0]
    mo = lookup[tok.convert_tokens_to_ids(object_token)]
    allow = cfg.model.allow_synthetic_weights
    unet_w, _, clip_w, synthetic = load_sd_weights(sd, str(cfg.model.pretrained_model_name_or_path), device, allow)
    dec_w, _ = load_vae_decoder_weights(sd, str(cfg.model.pretrained_model_name_or_path), device, allow)
    # grow the token table like Coach._extend_token_embedding (rows are placeholders for the mapper outputs)
    key = "text_model.embeddings.token_embedding.weight"
    E = clip_w[key]
    extra = len(tok) - E.shape[0]
    if extra > 0:
        so = tok.encode(cfg.data.super_category_object_token, add_special_tokens=False)[0]
        clip_w = dict(clip_w)
        clip_w[key] = torch.cat([E, E[so].unsqueeze(0).repeat(extra, 1)], 0)
    if height is None or width is None:
        if "dtu" in str(cfg.data.train_data_dir) and cfg.learnable_mode != 0:
            height, width = {0: (512, 512), 1: (384, 512), 2: (576, 768)}[cfg.data.dtu_preprocess_key]
        else:
            height = width = cfg.data.resolution
    m = cfg.model
    kw = {}
    if mapper_view is not None:
        kw = dict(mapper_view=mapper_view.mapper_state(), w_enc_view=mapper_view.encoder.w,
                  norm_scale_view=mapper_view.norm_scale, alpha_view=m.output_bypass_alpha_view,
                  unconstrained_view=m.bypass_unconstrained_view, output_bypass_view=mapper_view.output_bypass)
    eng = InferenceEngine(sd, unet_w, dec_w, clip_w, batch, height, width, mo.mapper_state(), mo.encoder.w,
                          mo.norm_scale, m.output_bypass_alpha_object, hidden_object=mo.hidden,
                          unconstrained_object=m.bypass_unconstrained_object, device=device,
                          **mo.engine_encoder_kwargs(), **kw)
    pipe = InferencePipeline(eng, tok, sampler)
    pm = PromptManager(tok, placeholder_view_token_ids=view_ids, placeholder_object_token_ids=object_ids,
                       view_params_fn=cam_fn)
    pipe.synthetic_weights = synthetic
    pipe.cfg = cfg
    return pipe, pm
