"""Seam B of INTEGRATION.md as real code: the module-call protocol `text_encoder(batch=NeTIBatch)` of the reference
(models/neti_clip_text_encoder.py:15-42,57-225; driven 16 times per step by `Coach.get_text_conditioning`,
training/coach.py:276-311, and T x 16 times per prompt by `PromptManager.embed_prompt`, prompt_manager.py:43-101) served
by the HIP text engine.

`HipNeTICLIPTextModel(batch=NeTIBatch)` returns the reference's pair `(output, output_with_bypass | None)`, each with
`.last_hidden_state` (B, 77, D) and `.pooler_output`; `[0]` indexing works like on `BaseModelOutputWithPooling`
(coach.py:296-304 reads `[0]`).  The reference calls the encoder once per UNet layer with `unet_layers` filled with that
layer's index; the engine computes ALL 16 layers' contexts of a (input_ids, placeholders, timesteps, truncation) batch in
one batched pass, so the first call of such a loop runs the kernels and the other fifteen read the cached result.
`text_model.embeddings.{set_mapper, mapper_object_lookup, mapper_view, token_embedding, position_embedding}`,
`get_input_embeddings()` and `resize_token_embeddings()` exist as on the reference module (coach.py:86-87,367,646-669).
Inference / evaluation only (no autograd graph): training runs through TrainStepEngine.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .. import lib
from .. import sd_config as sc
from ..engine.text import MapperState, TextEngine, flatten_mapper_state
from .checkpoint_handler import TextEncoderWeights
from .types import NeTIBatch


class _Output(tuple):
    """(last_hidden_state, pooler_output) with the attribute access of BaseModelOutputWithPooling"""

    def __new__(cls, last, pooled):
        o = super().__new__(cls, (last, pooled))
        o.last_hidden_state, o.pooler_output, o.hidden_states, o.attentions = last, pooled, None, None
        return o


class _Embeddings:
    def __init__(self, owner):
        self._owner = owner
        self.mapper_object_lookup: Optional[Dict[int, object]] = None
        self.mapper_view = None

    def set_mapper(self, mapper_object_lookup, mapper_view, device="cuda"):
        """models/net_clip_text_embedding.py:25-32"""
        self.mapper_object_lookup, self.mapper_view = mapper_object_lookup, mapper_view
        self._owner._engines.clear()

    @property
    def token_embedding(self):
        return SimpleNamespace(weight=self._owner.weights[TextEncoderWeights.KEY])

    @property
    def position_embedding(self):
        return SimpleNamespace(weight=self._owner.weights["text_model.embeddings.position_embedding.weight"])


class HipNeTICLIPTextModel(TextEncoderWeights):
    def __init__(self, clip_cfg: sc.CLIPTextConfig, clip_weights: Dict[str, torch.Tensor], n_unet_layers: int = 16,
                 device: str = "cuda"):
        super().__init__(dict(clip_weights))
        self.cfg, self.nl, self.dev = clip_cfg, n_unet_layers, device
        self.text_model = SimpleNamespace(embeddings=_Embeddings(self))
        self._engines: Dict[tuple, tuple] = {}
        self._cache_key, self._cache = None, None

    def resize_token_embeddings(self, n: int):
        self._engines.clear()
        return super().resize_token_embeddings(n)

    # ------------------------------------------------------------------ engine per (batch size, object mapper, view?)
    def _engine(self, B: int, obj_id: int, with_view: bool):
        key = (B, obj_id, with_view)
        if key not in self._engines:
            emb = self.text_model.embeddings
            mo_mod = emb.mapper_object_lookup[obj_id]
            D, L = self.cfg.hidden_size, self.cfg.max_positions
            ts = torch.zeros(B, dtype=torch.int64, device=self.dev)
            ck = torch.zeros((self.nl, B * L, D), dtype=lib.act_dtype(), device=self.dev)
            cv = torch.zeros_like(ck)
            w = mo_mod.encoder.w.to(self.dev).float().contiguous()
            mo = MapperState(flatten_mapper_state(mo_mod.mapper_state()).to(self.dev), None if mo_mod.legacy else w,
                             mo_mod.norm_scale, mo_mod.output_bypass_alpha, hidden=mo_mod.hidden, enc_dim=mo_mod.enc_dim,
                             unconstrained=mo_mod.bypass_unconstrained, legacy_w_pe=w if mo_mod.legacy else None,
                             output_bypass=mo_mod.output_bypass)
            mv = None
            if with_view:
                v = emb.mapper_view
                mv = MapperState(flatten_mapper_state(v.mapper_state()).to(self.dev),
                                 v.encoder.w.to(self.dev).float().contiguous(), v.norm_scale, v.output_bypass_alpha,
                                 unconstrained=v.bypass_unconstrained, output_bypass=v.output_bypass)
            eng = TextEngine(self.cfg, self.weights, self.nl, B, ts, ck, cv, None, None, mo, None, mv, None, 12, False,
                             self.dev, need_backward=False)
            eng.training = False
            eng.ensure_masks()
            self._engines[key] = (eng, ts, ck, cv, mo, mv)
        eng, ts, ck, cv, mo, mv = self._engines[key]
        # the engine computes from its OWN copy of the mapper parameters: refresh it on every (uncached) forward so an
        # optimizer step / load_state_dict on the modules is seen (108 k floats per mapper)
        emb = self.text_model.embeddings
        mo.params.copy_(flatten_mapper_state(emb.mapper_object_lookup[obj_id].mapper_state()))
        if mv is not None:
            mv.params.copy_(flatten_mapper_state(emb.mapper_view.mapper_state()))
        return eng, ts, ck, cv

    def _state_versions(self):
        """in-place edit counters of everything a forward depends on: every mapper tensor (object AND view) and the
        token / position tables (the Coach writes the placeholder rows in place, coach.py:367-395)"""
        emb = self.text_model.embeddings
        mods = list(emb.mapper_object_lookup.values()) + ([emb.mapper_view] if emb.mapper_view is not None else [])
        # (data_ptr, _version): mapper_state() returns fresh detached aliases on every call, whose id() is arbitrary and
        # reusable; an alias shares its base's storage pointer and version counter
        maps = tuple((int(t.data_ptr()), int(t._version)) for m in mods for t in m.mapper_state().values())
        tabs = tuple((int(self.weights[k].data_ptr()), int(self.weights[k]._version))
                     for k in (TextEncoderWeights.KEY, "text_model.embeddings.position_embedding.weight"))
        return maps, tabs

    @torch.no_grad()
    def __call__(self, input_ids: Optional[torch.Tensor] = None, batch: Optional[NeTIBatch] = None,
                 view_params: Optional[torch.Tensor] = None, **_):
        """batch: the reference's NeTIBatch.  view_params (B, 12) in [-1, 1]: the scaled camera parameters of the view
        tokens — the reference's view mapper looks them up from the token string itself (neti_mapper.py:265-337); here
        the caller does that once (Coach._view_params / PromptManager)."""
        emb = self.text_model.embeddings
        if batch is None:
            if input_ids is None:
                raise ValueError("You have to specify either batch or input_ids!")
            ids = input_ids.view(-1, input_ids.shape[-1])
            none = torch.full((ids.shape[0],), -1, dtype=torch.int64)
            batch = NeTIBatch(ids, none, none, torch.zeros(ids.shape[0], dtype=torch.int64),
                              torch.zeros(ids.shape[0], dtype=torch.int64))
            plain = True
        else:
            plain = False
        ids = batch.input_ids.view(-1, batch.input_ids.shape[-1]).cpu()
        B = ids.shape[0]
        ph_o, ph_v = batch.input_ids_placeholder_object.cpu(), batch.input_ids_placeholder_view.cpu()
        if emb.mapper_object_lookup is None:
            raise RuntimeError("set_mapper() first")
        if plain:
            obj_id = next(iter(emb.mapper_object_lookup))
        else:
            if not bool((ph_o == ph_o[0]).all()):
                raise AssertionError("a batch holds a single object token (net_clip_text_embedding.py:67-68)")
            obj_id = int(ph_o[0])
        with_view = emb.mapper_view is not None and not plain and not bool((ph_v == -1).all())
        layers = batch.unet_layers.cpu()
        if not bool((layers == layers[0]).all()):
            raise ValueError("unet_layers of one call hold one layer index (coach.py:289-295)")
        layer = int(layers[0])
        ts_host = batch.timesteps.cpu().to(torch.int64)
        maps, tabs = self._state_versions()
        if tabs != getattr(self, "_table_versions", tabs):
            self._engines.clear()  # the engines hold f16 device copies of the embedding tables
        self._table_versions = tabs
        key = (ids.numpy().tobytes(), ph_o.numpy().tobytes(), ph_v.numpy().tobytes(), ts_host.numpy().tobytes(),
               batch.truncation_idx, with_view, None if view_params is None else view_params.cpu().numpy().tobytes(),
               maps, tabs)
        if key != self._cache_key:
            eng, ts, ck, cv = self._engine(B, obj_id, with_view)
            if with_view and view_params is None:
                raise ValueError("view tokens in the batch: pass view_params (B, 12), scaled to [-1, 1]")
            eng.set_batch(ids, ph_o if not plain else torch.full((B,), -1), ph_v if with_view else None,
                          view_params if with_view else None)
            eng.set_truncation(batch.truncation_idx)
            ts.copy_(ts_host)
            eng.forward()
            L, D = self.cfg.max_positions, self.cfg.hidden_size
            self._cache = (ck.view(self.nl, B, L, D).float(), cv.view(self.nl, B, L, D).float())
            self._cache_key = key
        ck, cv = self._cache
        eot = ids.to(torch.int).argmax(dim=-1).to(ck.device)
        ar = torch.arange(B, device=ck.device)
        last = ck[layer]
        out = _Output(last, last[ar, eot])
        if plain:
            return out, None
        obj_mod = emb.mapper_object_lookup[obj_id]
        if not (obj_mod.output_bypass or (with_view and emb.mapper_view.output_bypass)):
            return out, None  # no mapper emits a bypass vector (neti_clip_text_encoder.py:207-225)
        last_b = cv[layer]
        return out, _Output(last_b, last_b[ar, eot])

    forward = __call__
