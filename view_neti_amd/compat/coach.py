"""Trainer with the surface of the reference's `training/coach.py::Coach` (`Coach(cfg).train()`), driving the
HIP train-step engine instead of diffusers/transformers/accelerate.

What is mirrored (file:line of the reference):
  * experiment directory, config.yaml, logs/log.txt                     training/logger.py:19-28
  * placeholder tokens appended to the tokenizer, their embedding rows initialised from the
    super-category rows, mapper norm_scale = |E[super]|                 coach.py:320-397
  * one object mapper per placeholder token (+ view mapper in modes 1-3), all initialised right after the
    encoder's torch.manual_seed(0) (App. C Q1)                           coach.py:492-598
  * learnable_mode 3: one object mapper per scene in one parameter bucket, the batch's scene picks the
    mapper (device-side slot), a new scene is drawn after every optimizer step   coach.py:155-156, dataset.py:584-596
  * nested dropout / unconstrained bypass flags forwarded to the mappers   coach.py:525-584
  * lr = lr * accum * batch * world when scale_lr                         coach.py:727-733
  * optim.lr_scheduler / lr_warmup_steps (compat/lr_schedule.py)          coach.py:759-770, :217
  * train loop, save every log.save_steps + final, file names            coach.py:137-274
  * validation images every eval.validation_steps (compat/validate.py: the live mappers on the inference engine;
    the DTU metric harness itself stays out of scope)                     coach.py:243-251, validate.py
What differs on purpose: DESIGN.md §5 (no embedding restore, device RNG, flat-bucket all-reduce).
"""
from __future__ import annotations

import logging
import sys
import time
from pathlib import Path
from typing import Dict, Optional

import torch

from .. import lib, parallel
from .. import sd_config as sc
from ..engine.step import TrainStepEngine
from ..engine.text import unflatten_mapper_state
from . import config as cfgmod
from .checkpoint_handler import CheckpointHandler
from .constants import UNET_LAYERS
from .dataset import TextualInversionDataset
from .lr_schedule import LRSchedule
from .neti_modules import NeTIMapper
from .sd_weights import load_sd_weights
from .tokenizer import load_tokenizer


def _sd_family(cfg) -> sc.SDConfig:
    """SD-1.x vs SD-2.x shape family from the word-embedding width (config.py:88-89)."""
    if cfg.model.word_embedding_dim == 1024:
        return sc.sd21()
    if cfg.model.word_embedding_dim == 768:
        return sc.sd15()
    if cfg.model.word_embedding_dim == sc.tiny().clip.hidden_size:
        return sc.tiny()
    raise ValueError(f"unsupported word_embedding_dim {cfg.model.word_embedding_dim}")


class Coach:
    def __init__(self, cfg: cfgmod.RunConfig, device: str = "cuda"):
        self.cfg = cfg
        self.rank, self.world, self.local_rank = parallel.world_info()
        self.device = device
        self._setup_logging()
        if cfg.optim.seed is not None:
            torch.manual_seed(cfg.optim.seed)
        if cfg.optim.mixed_precision not in ("fp16", "bf16"):
            # the reference default is "no" (fp32 everywhere, config.py:241); that path does not exist here and
            # running 16-bit under a config that says otherwise would be a silent change of numerics
            raise NotImplementedError(
                f"optim.mixed_precision='{cfg.optim.mixed_precision}': the HIP engine implements the two 16-bit branches of "
                "training/coach.py:792-802 (frozen UNet/VAE in f16 or bf16, f32 statistics and accumulation; the device "
                "GradScaler for fp16 only, as accelerate does); pass --optim.mixed_precision fp16 or bf16")
        # _get_weight_dtype (coach.py:792-802): the process computes in ONE 16-bit format — the fp16 or the bf16 build
        lib.set_precision(cfg.optim.mixed_precision)
        if cfg.optim.gradient_checkpointing:
            # accepted as a no-op: it trades memory for recompute without changing numerics, and the engine's saved
            # activations fit (13.5 GiB at bs=4)
            self.log("optim.gradient_checkpointing=True: ignored (the HIP engine keeps its activations; same numerics)")
        self.sd = _sd_family(cfg)
        self.tokenizer = load_tokenizer(str(cfg.model.pretrained_model_name_or_path), self.sd.clip.vocab_size)
        self.train_dataset = self._init_dataset()
        self._add_concept_tokens()
        unet_w, vae_w, clip_w, synthetic = load_sd_weights(self.sd, str(cfg.model.pretrained_model_name_or_path), device,
                                                           allow_synthetic=cfg.model.allow_synthetic_weights)
        self.synthetic_weights = synthetic
        if synthetic:
            self.log(f"WARNING: '{cfg.model.pretrained_model_name_or_path}' is not a local checkpoint directory; "
                     "training on SD-shaped SYNTHETIC weights (model.allow_synthetic_weights): checkpoints written by "
                     "this run are marked synthetic")
        clip_w = self._extend_token_embedding(clip_w)
        self.mapper_object_lookup, self.mapper_view = self._init_neti_mappers()
        # engine slot k <-> k-th placeholder object token (mapper_object_lookup, coach.py:505-552)
        self.object_slot = {tid: k for k, tid in enumerate(self.placeholder_object_token_ids)}
        objs = [self.mapper_object_lookup[tid] for tid in self.placeholder_object_token_ids] \
            if self.mapper_object_lookup is not None else [self._standin_object]
        first = objs[0]
        m = cfg.model
        bs = cfg.optim.train_batch_size
        lr = parallel.scaled_lr(cfg.optim.learning_rate, cfg.optim.gradient_accumulation_steps, bs, self.world,
                                cfg.optim.scale_lr)
        self.lr_schedule = LRSchedule(cfg.optim.lr_scheduler, lr, cfg.optim.lr_warmup_steps, cfg.optim.max_train_steps,
                                      cfg.optim.gradient_accumulation_steps, self.world)
        h, w = self._image_hw()
        kw = {}
        if self.mapper_view is not None:
            # modes 4/5: the loaded mapper carries the alpha it was trained with (checkpoint_handler.py:163-169, Q8)
            kw = dict(output_bypass_view=self.mapper_view.output_bypass,
                      mapper_view=self.mapper_view.mapper_state(), w_enc_view=self.mapper_view.encoder.w,
                      norm_scale_view=self.mapper_view.norm_scale, alpha_view=self.mapper_view.output_bypass_alpha,
                      train_view=cfg.learnable_mode != 5)
        self.engine = TrainStepEngine(
            self.sd, unet_w, vae_w, clip_w, bs, h, w, [o.mapper_state() for o in objs], first.encoder.w,
            first.norm_scale, m.output_bypass_alpha_object, lr=lr, betas=(cfg.optim.adam_beta1, cfg.optim.adam_beta2),
            adam_eps=cfg.optim.adam_epsilon, weight_decay=cfg.optim.adam_weight_decay,
            seed=parallel.data_seed(cfg.seed, self.rank), world_size=self.world, device=device,
            grad_accum=cfg.optim.gradient_accumulation_steps, hidden_object=first.hidden,
            unconstrained_object=m.bypass_unconstrained_object, unconstrained_view=m.bypass_unconstrained_view,
            nested_dropout_prob=m.nested_dropout_prob if m.use_nested_dropout else 0.0,
            moment_cache_images=self._moment_cache_size(), **first.engine_encoder_kwargs(), **kw)
        self.engine.set_lr(self.lr_schedule.lr(0))
        self.validator = None
        if cfg.eval.validation_prompts is not None and cfg.eval.validation_steps <= cfg.optim.max_train_steps \
                and self.rank == 0:
            from .sd_weights import load_vae_decoder_weights
            from .validate import ValidationHandler
            dec_w, _ = load_vae_decoder_weights(self.sd, str(cfg.model.pretrained_model_name_or_path), device,
                                                allow_synthetic=cfg.model.allow_synthetic_weights)
            self.validator = ValidationHandler(self, unet_w, dec_w, clip_w)
        del unet_w, vae_w, clip_w
        self.checkpoint_handler = CheckpointHandler(
            cfg, self.train_dataset.placeholder_view_tokens, self.placeholder_view_token_ids,
            self.train_dataset.placeholder_object_tokens, self.placeholder_object_token_ids, cfg.log.exp_dir,
            synthetic_weights=synthetic)
        # every rank draws its own batches (accelerate shards the prepared dataloader across processes,
        # coach.py:97-99): one shuffling stream per rank; rank 0 of a 1-process run keeps the global generator
        gen = None
        if self.world > 1:
            gen = torch.Generator()
            gen.manual_seed(parallel.data_seed(cfg.seed, self.rank))
        self.train_dataloader = torch.utils.data.DataLoader(self.train_dataset, batch_size=bs, shuffle=True,
                                                            num_workers=cfg.data.dataloader_num_workers,
                                                            drop_last=True, generator=gen,
                                                            collate_fn=TextualInversionDataset.collate)
        self.device_pipe, self._device_sources = None, {}
        if getattr(cfg.data, "device_input_pipeline", False):
            from ..engine.input_pipeline import DeviceImagePipeline
            _, _, ph, pw = self.engine.pixel_values.shape
            self.device_pipe = DeviceImagePipeline(ph, pw, device)

    def _moment_cache_size(self) -> int:
        """`data.cache_vae_moments` (extension): legal only where the pixels of dataset item i are the same
        every time it comes up — augmentation_key 0 (no random crop / jitter / blur / rotation; flips are never enabled,
        coach.py:682-702) and a flat image list (not the per-scene dict of learnable_mode 3)."""
        d = self.cfg.data
        if not getattr(d, "cache_vae_moments", False):
            return 0
        ds = self.train_dataset
        if d.augmentation_key != 0 or not isinstance(ds.image_paths, (list, tuple)):
            raise ValueError("data.cache_vae_moments needs a deterministic dataset: augmentation_key 0 and a single image "
                             f"list (augmentation_key {d.augmentation_key}, learnable_mode {self.cfg.learnable_mode})")
        return int(ds.num_images)

    def _pixels(self, batch):
        """host path: the collated f32 batch; device path (cfg.data.device_input_pipeline): the plans drawn by the
        dataset are executed by HIP kernels straight into the engine's pixel buffer (returns None = already there)"""
        if "aug" not in batch:
            return batch["pixel_values"]
        ds, pipe = self.train_dataset, self.device_pipe
        for b, a in enumerate(batch["aug"]):
            src = self._device_sources.get(a["path"])
            if src is None:
                src = self._device_sources[a["path"]] = pipe.upload(ds.load_source(a["path"]))
            pipe.run(src, self.engine.pixel_values[b], resize=ds.target_size(), flip=a["flip"], plan=a["plan"])
        return None

    # ------------------------------------------------------------------ set-up
    def _setup_logging(self):
        cfg = self.cfg
        self.logger = logging.getLogger(f"vneti.coach.{id(self)}")
        self.logger.setLevel(logging.INFO)
        fmt = logging.Formatter("%(asctime)s %(message)s", "%Y-%m-%d %H:%M:%S")
        if self.rank == 0:
            cfg.log.exp_dir.mkdir(parents=True, exist_ok=True)
            cfg.log.logging_dir.mkdir(parents=True, exist_ok=True)
            for h in (logging.StreamHandler(sys.stdout), logging.FileHandler(cfg.log.logging_dir / "log.txt")):
                h.setFormatter(fmt)
                self.logger.addHandler(h)
            with (cfg.log.exp_dir / "config.yaml").open("w") as f:
                cfgmod.dump(cfg, f)

    def log(self, msg: str):
        if self.rank == 0:
            self.logger.info(msg)

    def _image_hw(self):
        """(H, W) of the tensors the dataset really produces (`_resize`, dataset.py:155-236): DTU roots give
        384x512 / 576x768 frames in EVERY learnable mode, LLFF roots are not resized at all."""
        hw = self.train_dataset.target_size()
        if hw is None:
            raise NotImplementedError("this data root is not resized by the dataset (llff): the engine needs one "
                                      "static frame size")
        return hw

    def _init_dataset(self):
        d = self.cfg.data
        return TextualInversionDataset(
            data_root=d.train_data_dir, tokenizer=self.tokenizer, size=d.resolution,
            placeholder_object_token=d.placeholder_object_token, repeats=d.repeats, center_crop=d.center_crop,
            set="train", learnable_mode=self.cfg.learnable_mode, camera_representation=d.camera_representation,
            train_data_subsets=d.train_data_subsets, placeholder_object_tokens=d.placeholder_object_tokens,
            fixed_object_token_or_path=d.fixed_object_token_or_path, dtu_lighting=d.dtu_lighting,
            dtu_subset=d.dtu_subset, caption_strategy=d.caption_strategy, dtu_preprocess_key=d.dtu_preprocess_key,
            augmentation_key=d.augmentation_key,  # flip_p is NOT forwarded — reference quirk Q10
            device_pipeline=bool(getattr(d, "device_input_pipeline", False)))

    def _add_concept_tokens(self):
        ds, tok = self.train_dataset, self.tokenizer
        self.cfg.data.placeholder_view_tokens = list(ds.placeholder_view_tokens)
        n = tok.add_tokens(ds.placeholder_view_tokens + ds.placeholder_object_tokens)
        if n == 0:
            raise ValueError("No new tokens were added to the tokenizer")
        self.placeholder_view_token_ids = tok.convert_tokens_to_ids(ds.placeholder_view_tokens) \
            if ds.placeholder_view_tokens else []
        self.placeholder_object_token_ids = tok.convert_tokens_to_ids(ds.placeholder_object_tokens)
        enc = lambda t: tok.encode(t, add_special_tokens=False)
        so, sv = enc(self.cfg.data.super_category_object_token), enc(self.cfg.data.super_category_view_token)
        if len(so) != 1 or len(sv) != 1:
            raise ValueError("super-category tokens must be single vocabulary tokens")
        self.super_object_id, self.super_view_id = so[0], sv[0]

    def _extend_token_embedding(self, clip_w):
        key = "text_model.embeddings.token_embedding.weight"
        E = clip_w[key]
        extra = len(self.tokenizer) - E.shape[0]
        rows = torch.empty(extra, E.shape[1], dtype=E.dtype, device=E.device)
        base = E.shape[0]
        for i in self.placeholder_view_token_ids:
            rows[i - base] = E[self.super_view_id]
        for i in self.placeholder_object_token_ids:
            rows[i - base] = E[self.super_object_id]
        clip_w = dict(clip_w)
        clip_w[key] = torch.cat([E, rows], 0)
        m = self.cfg.model
        m.target_norm_view = float(E[self.super_view_id].norm()) if m.normalize_view_mapper_output else None
        m.target_norm_object = float(E[self.super_object_id].norm()) if m.normalize_object_mapper_output else None
        self.token_embedding = clip_w[key]
        return clip_w

    def _init_neti_mappers(self):
        cfg, m = self.cfg, self.cfg.model
        legacy = m.arch_view_net <= 14
        if legacy:
            # the dataclass default (config.py:130): NeTIPositionalEncoding + anchor-initialised input_layer; the view
            # side of that era needs `encode_phi`, which the reference never defines (neti_mapper.py:347-348)
            if cfg.learnable_mode != 0:
                raise NotImplementedError("arch_view_net <= 14 (legacy mappers) works for object mappers only "
                                          "(learnable_mode 0); the view modes need --model.arch_view_net 15 "
                                          "--model.arch_view_disable_tl False")
            if int(m.use_positional_encoding_object) != 1:
                raise NotImplementedError("legacy mapper: only use_positional_encoding_object = 1 (NeTIPositionalEncoding)")
        elif m.arch_view_net != 15 or m.arch_view_disable_tl:
            raise NotImplementedError("the HIP engine implements arch_view_net = 15 (with arch_view_disable_tl False, "
                                      "neti_mapper.py:481-483) and the legacy object mapper of arch_view_net <= 14")
        if m.original_ti:
            # not "unsupported here" but broken upstream: with original_ti the mapper returns a bare tensor and
            # NeTICLIPTextEmbeddings.forward reads `.output_bypass_alpha` off it (net_clip_text_embedding.py:80) —
            # AttributeError in the reference itself (probed with the real modules)
            raise NotImplementedError("original_ti: the reference's own path raises AttributeError "
                                      "(net_clip_text_embedding.py:80 on the tensor neti_mapper.py:183-192 returns)")
        if cfg.learnable_mode == 1 and Path(str(cfg.data.fixed_object_token_or_path)).exists():
            # Upstream this combination is unfinished: coach.py:554-557 loads the pretrained object mapper into a LOCAL that is
            # never used again, `mapper_object_lookup` stays None in mode 1 (:493,508; "todo: option to keep learning a
            # pretrained object mapper", :669), so NeTICLIPTextEmbeddings embeds the placeholder token through its plain
            # (super-category-initialised) table row (net_clip_text_embedding.py:65) — the checkpoint is silently ignored.
            # Reproducing that would train a view mapper against an object the user did not ask for: refuse instead.
            raise NotImplementedError("learnable_mode 1 with a pretrained object mapper (fixed_object_token_or_path is a "
                                      ".pt file): the reference itself loads that mapper and never uses it "
                                      "(training/coach.py:554-557, mapper_object_lookup stays None); pass a vocabulary "
                                      "word, or train object and view mapper together with learnable_mode 2")
        if (m.bypass_unconstrained_object and not m.output_bypass_object) or \
                (m.bypass_unconstrained_view and not m.output_bypass_view):
            raise ValueError("bypass_unconstrained needs output_bypass (neti_mapper.py:130-132)")
        if len(UNET_LAYERS) != self.sd.unet.n_cross_layers:
            raise ValueError("UNET_LAYERS does not match the UNet")
        lookup = {}
        for token, token_id in zip(self.train_dataset.placeholder_object_tokens, self.placeholder_object_token_ids):
            lookup[token_id] = NeTIMapper("object", m.word_embedding_dim, m.arch_mlp_hidden_dims, m.target_norm_object,
                                          m.pe_sigmas, m.output_bypass_object, m.bypass_unconstrained_object,
                                          m.output_bypass_alpha_object, token,
                                          use_nested_dropout=m.use_nested_dropout,
                                          nested_dropout_prob=m.nested_dropout_prob, arch_view_net=m.arch_view_net,
                                          num_pe_time_anchors=m.num_pe_time_anchors)
        view = None
        if cfg.learnable_mode in (1, 2, 3):
            ds = self.train_dataset
            cams = torch.stack(list(ds.lookup_camidx_to_cam_params.values()))
            view = NeTIMapper("view", m.word_embedding_dim, 64, m.target_norm_view, m.pe_sigmas, m.output_bypass_view,
                              m.bypass_unconstrained_view, m.output_bypass_alpha_view, None,
                              cams.min(0).values.flatten(), cams.max(0).values.flatten(),
                              use_nested_dropout=m.use_nested_dropout, nested_dropout_prob=m.nested_dropout_prob)
        elif cfg.learnable_mode in (4, 5):
            ds = self.train_dataset
            cams = torch.stack(list(ds.lookup_camidx_to_cam_params.values()))
            _, view = CheckpointHandler.load_mapper(m.pretrained_view_mapper, "view", cam_mins=cams.min(0).values.flatten(),
                                                    cam_maxs=cams.max(0).values.flatten())
        if cfg.learnable_mode == 1:
            # view mapper only; the object is a plain vocabulary word (dataset.py:654-668: no placeholder id in the batch,
            # mapper_object_lookup stays None, coach.py:493,508).  The engine's bucket layout wants an object segment: it
            # gets a stand-in mapper that no prompt ever reaches (placeholder -1 never matches a position) and that is
            # neither registered nor saved.
            self._standin_object = NeTIMapper("object", m.word_embedding_dim, m.arch_mlp_hidden_dims, m.target_norm_object,
                                              m.pe_sigmas, m.output_bypass_object, m.bypass_unconstrained_object,
                                              m.output_bypass_alpha_object, None, arch_view_net=m.arch_view_net,
                                              num_pe_time_anchors=m.num_pe_time_anchors)
            return None, view
        if cfg.learnable_mode != 3 and len(lookup) != 1:
            raise ValueError("only learnable_mode 3 trains more than one object token (dataset.py:612)")
        return lookup, view

    # ------------------------------------------------------------------ training
    def _view_params(self, ids_view: torch.Tensor) -> Optional[torch.Tensor]:
        """camera parameters are parsed back out of the token STRING (4-decimal quantisation, Q15) and scaled
        to [-1,1] with the min/max over all calibration files (neti_mapper.py:265-337)."""
        if self.mapper_view is None:
            return None
        ds, mv = self.train_dataset, self.mapper_view
        id2tok = dict(zip(self.placeholder_view_token_ids, ds.placeholder_view_tokens))
        p = torch.stack([ds.dtu_token_to_cam_params(id2tok[int(i)])[0] for i in ids_view])
        return (p - mv.cam_mins) / (mv.cam_maxs - mv.cam_mins) * 2 - 1

    def _sync_modules(self):
        """copy the trained flat bucket back into the nn.Module views used for checkpoints."""
        eng, D = self.engine, self.cfg.model.word_embedding_dim
        for tid, k in self.object_slot.items():
            obj = self.mapper_object_lookup[tid]
            od = 2 * D if obj.output_bypass else D
            obj.load_state_dict(unflatten_mapper_state(eng.object_params(k).cpu(), obj.enc_dim, obj.hidden, od,
                                                       obj.pe_dim), strict=False)
        if self.mapper_view is not None and eng.view_params_flat().numel() > 0:
            self.mapper_view.load_state_dict(
                unflatten_mapper_state(eng.view_params_flat().cpu(), 64, 64, 2 * D if self.mapper_view.output_bypass else D),
                strict=False)

    def save(self, embeds_name: str, mapper_name: str):
        if self.rank != 0:
            return
        self._sync_modules()
        self.checkpoint_handler.save_model(self.token_embedding, self.mapper_object_lookup, self.mapper_view,
                                           embeds_name, mapper_name)

    def train(self):
        cfg, eng = self.cfg, self.engine
        total_bs = cfg.optim.train_batch_size * self.world * cfg.optim.gradient_accumulation_steps
        for line in ("***** Running training *****", f"  Num examples = {len(self.train_dataset)}",
                     f"  Instantaneous batch size per device = {cfg.optim.train_batch_size}",
                     f"  Total train batch size (w. parallel, distributed & accumulation) = {total_bs}",
                     f"  Gradient Accumulation steps = {cfg.optim.gradient_accumulation_steps}",
                     f"  Total optimization steps = {cfg.optim.max_train_steps}"):
            self.log(line)
        global_step, captured, t0 = 0, False, time.time()
        if cfg.learnable_mode == 3:
            # every rank must draw the same scene each step (one all-reduce of that scene's mapper); the
            # reference leaves np.random unseeded per process, which only works because its DDP wrapper never
            # sees the dict-held object mappers
            import numpy as np
            np.random.seed(cfg.seed if cfg.seed is not None else 0)
        while global_step < cfg.optim.max_train_steps:
            for batch in self.train_dataloader:
                ids_obj = batch["input_ids_placeholder_object"]
                if not bool((ids_obj == ids_obj[0]).all()):
                    raise ValueError("a batch must hold a single object token (net_clip_text_embedding.py:67-68)")
                eng.set_batch(self._pixels(batch), batch["input_ids"], ids_obj, batch["input_ids_placeholder_view"],
                              self._view_params(batch["input_ids_placeholder_view"]),
                              object_index=self.object_slot.get(int(ids_obj[0]), 0),  # (-1 in mode 1: no object mapper)
                              image_idx=batch["image_idx"] if eng.n_cache else None)
                if not captured:
                    eng.capture()
                    captured = True
                stepped = eng.step()
                if stepped and cfg.learnable_mode == 3:
                    self.train_dataset.reset_sampled_object()  # new scene only once the accumulation group is done
                if stepped:
                    global_step += 1
                    if not self.lr_schedule.constant:
                        eng.set_lr(self.lr_schedule.lr(global_step))  # device scalar: no re-capture
                    if global_step % 50 == 0 or global_step == 1:
                        self.log(f"step {global_step} loss {eng.loss():.5f} lr {float(eng.hyper[0]):.2e} "
                                 f"{global_step / (time.time() - t0):.2f} it/s")
                    if global_step % cfg.log.save_steps == 0:
                        self.save(f"learned_embeds-steps-{global_step}.bin", f"mapper-steps-{global_step}.pt")
                    if self.validator is not None and global_step % cfg.eval.validation_steps == 0:  # coach.py:243,834
                        self.validator.infer(global_step)
                if global_step >= cfg.optim.max_train_steps:
                    break
        torch.cuda.synchronize()
        self.save("learned_embeds-final.bin", "mapper-final.pt")
