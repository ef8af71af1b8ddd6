"""Mapper / learned-embedding checkpoints in the reference's on-disk format (checkpoint_handler.py:34-267,
SURVEY App. D), written from and read into the HIP engine's flat parameter bucket.

  learned_embeds-*.bin      torch.save({token_str: Tensor(D,) cpu fp32}), view tokens first, then object
  mapper-*_object.pt        {"cfg": encode(RunConfig), "mappers": {token_id: {"state_dict", "encoder",
                                                                         "placeholder_object_token"}}}
  mapper-*_view.pt          same with the single key "dummy_key" / token "dummy"
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import torch

from . import config as cfgmod
from .neti_modules import NeTIMapper


class TextEncoderWeights:
    """The slice of the reference's `text_encoder` module protocol that `load_learned_embed_in_clip` touches
    (`get_input_embeddings().weight`, `resize_token_embeddings(n)`, checkpoint_handler.py:241,256,264), over the CLIP
    weight dict the HIP engines consume (key `text_model.embeddings.token_embedding.weight`)."""
    KEY = "text_model.embeddings.token_embedding.weight"

    class _Emb:
        def __init__(self, owner):
            self._owner = owner

        @property
        def weight(self):
            return self._owner.weights[TextEncoderWeights.KEY]

    def __init__(self, clip_weights: Dict[str, torch.Tensor]):
        self.weights = clip_weights

    def get_input_embeddings(self):
        return TextEncoderWeights._Emb(self)

    def resize_token_embeddings(self, n: int):
        E = self.weights[self.KEY]
        if n > E.shape[0]:  # new rows: every one of them is assigned by the caller right after
            self.weights[self.KEY] = torch.cat([E, torch.zeros(n - E.shape[0], E.shape[1], dtype=E.dtype,
                                                               device=E.device)], 0)
        elif n < E.shape[0]:
            self.weights[self.KEY] = E[:n].clone()
        return self.get_input_embeddings()


class CheckpointHandler:
    def __init__(self, cfg, placeholder_view_tokens: List[str], placeholder_view_token_ids: List[int],
                 placeholder_object_tokens: List[str], placeholder_object_token_ids: List[int], save_root: Path,
                 synthetic_weights: bool = False):
        self.cfg = cfg
        self.synthetic_weights = synthetic_weights
        self.placeholder_tokens = list(placeholder_view_tokens) + list(placeholder_object_tokens)
        self.placeholder_token_ids = list(placeholder_view_token_ids) + list(placeholder_object_token_ids)
        self.save_root = Path(save_root)

    def save_model(self, token_embedding: torch.Tensor, mapper_object_lookup: Optional[Dict[int, NeTIMapper]],
                   mapper_view: Optional[NeTIMapper], embeds_save_name: str, mapper_save_name: str):
        self.save_learned_embeds(token_embedding, embeds_save_name)
        self.save_mapper(mapper_object_lookup, mapper_view, mapper_save_name)

    def save_learned_embeds(self, token_embedding: torch.Tensor, save_name: str):
        rows = token_embedding[self.placeholder_token_ids].detach().float().cpu()
        torch.save({t: v.clone() for t, v in zip(self.placeholder_tokens, rows)}, self.save_root / save_name)

    def save_mapper(self, mapper_object_lookup, mapper_view, save_name: str):
        # "cfg" holds the reference's schema only (its pyrallis.decode rejects unknown keys, checkpoint_handler.py:142);
        # this repo's extension fields and the synthetic-weights marker travel under a separate top-level key that the
        # reference's loader never looks at
        enc_cfg = cfgmod.encode(self.cfg, include_ext=False)
        ext = {"config_ext": cfgmod.ext_fields(self.cfg), "synthetic_sd_weights": bool(self.synthetic_weights)}
        stem, suffix = Path(save_name).stem, Path(save_name).suffix
        if mapper_object_lookup is not None:
            sd = {"cfg": enc_cfg, "vneti_ext": ext, "mappers": {}}
            for token_id, m in mapper_object_lookup.items():
                sd["mappers"][token_id] = {"state_dict": m.mapper_state(), "encoder": m.encoder,
                                           "placeholder_object_token": m.placeholder_object_token}
            torch.save(sd, os.path.join(self.save_root, stem + "_object" + suffix))
        if mapper_view is not None:
            sd = {"cfg": enc_cfg, "vneti_ext": ext,
                  "mappers": {"dummy_key": {"state_dict": mapper_view.mapper_state(),
                                                            "encoder": mapper_view.encoder,
                                                            "placeholder_object_token": "dummy"}}}
            torch.save(sd, os.path.join(self.save_root, stem + "_view" + suffix))

    @staticmethod
    def clean_config_dict(d):
        """drop the None-valued run-time fields that would not decode (checkpoint_handler.py:99-127)."""
        d["data"].pop("placeholder_view_tokens", None)
        for sec, keys in (("model", ["target_norm_object", "target_norm_view", "pretrained_view_mapper",
                                     "pretrained_view_mapper_key"]),
                          ("eval", ["validation_view_tokens", "eval_placeholder_object_tokens"]),
                          ("data", ["placeholder_object_tokens", "train_data_subsets"])):
            for k in keys:
                if k in d.get(sec, {}) and d[sec][k] is None:
                    del d[sec][k]
        return d

    @staticmethod
    def load_mapper(mapper_path: Path, embedding_type: str = "object", placeholder_object_tokens: List[str] = None,
                    placeholder_object_token_ids: List[int] = None, cam_mins=None, cam_maxs=None
                    ) -> Tuple[object, object]:
        """-> (RunConfig, {token_id: NeTIMapper}) for objects, (RunConfig, NeTIMapper) for the view mapper."""
        ckpt = torch.load(mapper_path, map_location="cpu", weights_only=False)
        raw = ckpt["cfg"]
        cfg = cfgmod.decode(cfgmod.RunConfig, CheckpointHandler.clean_config_dict(dict(raw)))
        mc = cfg.model
        is_view = embedding_type == "view"
        target_norm = mc.target_norm_view if is_view else mc.target_norm_object
        if not is_view and target_norm is None and mc.normalize_object_mapper_output:
            raise ValueError("need a target norm to pass to pretrained object mapper")
        # quirk kept (App. C Q8): the object alpha is read for both mapper kinds
        alpha = raw["model"].get("output_bypass_alpha_object", 0.2)
        unconstrained = raw["model"].get("bypass_unconstrained_view" if is_view else "bypass_unconstrained_object", False)
        out = {}
        for key, entry in ckpt["mappers"].items():
            m = NeTIMapper(embedding_type=embedding_type, output_dim=mc.word_embedding_dim,
                           arch_mlp_hidden_dims=mc.arch_mlp_hidden_dims, norm_scale=target_norm,
                           pe_sigmas=mc.pe_sigmas, output_bypass=mc.output_bypass_view if is_view else mc.output_bypass_object,
                           bypass_unconstrained=unconstrained, output_bypass_alpha=alpha,
                           placeholder_object_token=entry["placeholder_object_token"], cam_mins=cam_mins,
                           cam_maxs=cam_maxs, use_nested_dropout=mc.use_nested_dropout,
                           nested_dropout_prob=mc.nested_dropout_prob, arch_view_net=mc.arch_view_net,
                           num_pe_time_anchors=mc.num_pe_time_anchors)
            if m.legacy:
                # the legacy frequencies are NOT seeded: the pickled encoder instance is their only record
                # (checkpoint_handler.py:213-215 copies them over the freshly drawn ones as well)
                m.encoder.w = entry["encoder"].w.detach().float().cpu().clone()
            state = dict(entry["state_dict"])
            missing = set(m.mapper_state()) ^ set(state)
            if missing:
                raise RuntimeError(f"mapper state_dict keys differ: {sorted(missing)}")
            m.load_state_dict(state, strict=False)  # strict on the mapper keys (checked above); encoder.w is regenerated
            m.eval()
            if is_view:
                return cfg, m
            lookup = dict(zip(placeholder_object_tokens, placeholder_object_token_ids))
            out[lookup[entry["placeholder_object_token"]]] = m
        return cfg, out

    @staticmethod
    def load_learned_embed_in_clip(learned_embeds_path: Path, text_encoder, tokenizer) -> Tuple[List[str], List[int]]:
        """checkpoint_handler.py:232-267: add the saved placeholder tokens to the tokenizer, grow the token-embedding
        table and write the saved rows into it.  `text_encoder` is anything with the two methods the reference calls
        (`TextEncoderWeights` wraps the engines' weight dict).  -> (tokens, token ids)"""
        loaded = torch.load(learned_embeds_path, map_location="cpu")
        trained_tokens = list(loaded.keys())
        dtype = text_encoder.get_input_embeddings().weight.dtype
        embeds = [e.to(dtype) for e in loaded.values()]
        if tokenizer.add_tokens(trained_tokens) == 0:
            raise ValueError(f"The tokenizer already contains the token {trained_tokens[0]}. "
                             f"Please pass a different `token` that is not already in the tokenizer.")
        text_encoder.resize_token_embeddings(len(tokenizer))
        ids = [tokenizer.convert_tokens_to_ids(t) for t in trained_tokens]
        table = text_encoder.get_input_embeddings().weight
        for token_id, embed in zip(ids, embeds):
            table.data[token_id] = embed.to(table.device)
        return trained_tokens, ids
