"""Validation images during training — the trigger and file naming of the reference's `ValidationHandler`
(training/validate.py, called from coach.py:243-251 when `global_step % eval.validation_steps == 0`), without its
DTU metric harness (masked MSE/PSNR/SSIM/LPIPS against ground-truth views needs the DTU masks and LPIPS weights:
SURVEY §8 f4, out of scope).

What is produced, with the live mapper parameters (the inference engine aliases the trainer's parameter bucket):
  * learnable_mode 0: every `eval.validation_prompts` template filled with the object token x `eval.validation_seeds`
    -> one grid per prompt, `validation-iter_{step}-…_imgs_t2i_{i}.png` (validate.py:296-309);
  * view modes: "<view token>. A photo of a <object>" for the training views x seeds -> the camidx -> images dict the
    reference saves (`validation-iter_{step}-denoisesteps_{N}_numseeds_{k}_upsample_{u}.pt`, validate.py:111-118)
    and a grid PNG per seed.
"""
from __future__ import annotations

from pathlib import Path
from typing import List

import numpy as np
import torch
from PIL import Image

from ..engine.infer import InferenceEngine
from .prompt_manager import PromptManager
from .sd_pipeline_call import InferencePipeline, sd_pipeline_call


def make_grid(images: List[np.ndarray], nrow: int) -> Image.Image:
    """uint8 HWC images -> one image, row-major, `nrow` per row (torchvision.utils.make_grid without padding)."""
    h, w = images[0].shape[:2]
    rows = (len(images) + nrow - 1) // nrow
    canvas = np.zeros((rows * h, nrow * w, 3), dtype=np.uint8)
    for i, im in enumerate(images):
        r, c = divmod(i, nrow)
        canvas[r * h:(r + 1) * h, c * w:(c + 1) * w] = im
    return Image.fromarray(canvas)


class ValidationHandler:
    def __init__(self, coach, unet_w, vae_dec_w, clip_w):
        self.coach = coach
        self.cfg = cfg = coach.cfg
        eng = coach.engine
        m = cfg.model
        # learnable_mode 1 trains no object mapper (the object is a vocabulary word, dataset.py:654-668): the engine's object
        # side is the Coach's stand-in, which no prompt reaches (placeholder -1)
        first = coach._standin_object if cfg.learnable_mode == 1 else \
            coach.mapper_object_lookup[coach.placeholder_object_token_ids[0]]
        h, w = coach._image_hw()
        kw = {}
        if coach.mapper_view is not None:
            mv = coach.mapper_view
            frozen = eng.view_params_flat().numel() == 0  # modes 4/5 train no view mapper: it is not in the bucket
            kw = dict(w_enc_view=mv.encoder.w, norm_scale_view=mv.norm_scale, alpha_view=m.output_bypass_alpha_view,
                      unconstrained_view=m.bypass_unconstrained_view, output_bypass_view=mv.output_bypass,
                      **(dict(mapper_view=mv.mapper_state()) if frozen else dict(params_view=eng.view_params_flat())))
        self.slot = torch.zeros(1, dtype=torch.int32, device=eng.dev)
        self.engine = InferenceEngine(
            coach.sd, unet_w, vae_dec_w, clip_w, 1, h, w, None, first.encoder.w, first.norm_scale,
            m.output_bypass_alpha_object, hidden_object=first.hidden, unconstrained_object=m.bypass_unconstrained_object,
            device=eng.dev, params_object=eng.params[: eng.n_all_obj], object_slot=self.slot,
            object_slot_stride=eng.n_obj, **first.engine_encoder_kwargs(), **kw)
        self.pipeline = InferencePipeline(self.engine, coach.tokenizer, "dpm++2m")
        self.prompt_manager = PromptManager(
            coach.tokenizer, placeholder_view_token_ids=coach.placeholder_view_token_ids,
            placeholder_object_token_ids=coach.placeholder_object_token_ids,
            view_params_fn=lambda tid: coach._view_params(torch.tensor([tid]))[0])

    def _generate(self, prompt: str, seeds: List[int]) -> List[np.ndarray]:
        emb = self.prompt_manager.embed_prompt(prompt)
        tid = int(emb.input_ids_placeholder_object)
        self.slot.fill_(self.coach.object_slot.get(tid, 0))
        out = []
        for seed in seeds:
            img = sd_pipeline_call(self.pipeline, emb, num_inference_steps=self.cfg.eval.num_denoising_steps,
                                   generator=torch.Generator().manual_seed(seed), output_type="np",
                                   return_dict=False)[0]
            out.append((img[0] * 255).round().astype(np.uint8))
        return out

    def infer(self, step: int):
        cfg, coach = self.cfg, self.coach
        ev = cfg.eval
        exp = Path(cfg.log.exp_dir)
        stem = f"validation-iter_{step}-denoisesteps_{ev.num_denoising_steps}"
        seeds = list(ev.validation_seeds)
        if cfg.learnable_mode == 0:
            token = coach.train_dataset.placeholder_object_tokens[0]
            for i, tmpl in enumerate(ev.validation_prompts or []):
                imgs = self._generate(tmpl.format(token), seeds)
                make_grid(imgs, len(imgs)).save(exp / f"{stem}_upsample_{ev.dtu_upsample_key}_imgs_t2i_{i}.png")
            return
        ds = coach.train_dataset
        if cfg.learnable_mode == 1:
            tokens = [ds.fixed_object_token]  # validate.py:455: modes 1, 2, 4, 5 share the view-token prompts; here with the word
        else:
            tokens = (ev.eval_placeholder_object_tokens or ds.placeholder_object_tokens[:1]) if cfg.learnable_mode == 3 \
                else ds.placeholder_object_tokens[:1]
        result = {}
        for obj in tokens:
            per_cam = {}
            for view_token in ds.placeholder_view_tokens:
                cam = ds.lookup_view_token_to_camidx[view_token]
                per_cam[cam] = self._generate(f"{view_token}. A photo of a {obj}", seeds)
            result[obj] = per_cam
            for si, seed in enumerate(seeds):
                grid = make_grid([per_cam[c][si] for c in sorted(per_cam)], min(len(per_cam), 7))
                tag = "" if len(tokens) == 1 else "_" + obj.strip("<>")
                grid.save(exp / f"{stem}_numseeds_{len(seeds)}_upsample_{ev.dtu_upsample_key}{tag}_seed_{seed}.png")
        torch.save(result[tokens[0]] if len(tokens) == 1 else result,
                   exp / f"{stem}_numseeds_{len(seeds)}_upsample_{ev.dtu_upsample_key}.pt")
        if cfg.data.camera_representation == "dtu-12d":
            self._dtu_metrics(step, stem, result, seeds)

    def _dtu_metrics(self, step: int, stem: str, result, seeds):
        """validate.py:123-186: masked MSE / PSNR / SSIM of the generated views against the scene's ground truth
        (compat/dtu_metrics.py).  Runs when all the evaluation views of the split were generated at a 3:4 aspect and the
        ground-truth images are on disk; object masks default to all-white when the IDR masks are absent."""
        import json
        from . import dtu_metrics as dm
        cfg, coach = self.cfg, self.coach
        cam_idxs, _, _ = dm.get_cam_idxs(cfg.data.dtu_subset)
        summary = {}
        for obj, per_cam in result.items():
            if set(per_cam) != set(cam_idxs):
                continue
            h, w = per_cam[cam_idxs[0]][0].shape[:2]
            if h / w != 0.75:
                continue
            if cfg.learnable_mode == 3:
                scan_id = obj[5:-1]
                scene = Path(cfg.data.train_data_dir) / f"scan{scan_id}"
            else:
                scene, scan_id = Path(cfg.data.train_data_dir), Path(cfg.data.train_data_dir).stem[4:]
            if not scene.exists():
                continue
            pred = {c: np.stack(per_cam[c]) for c in cam_idxs}
            try:
                res = dm.evaluate_dtu_predictions(pred, scene, cfg.data.dtu_subset, cfg.data.dtu_lighting, 1, seeds,
                                                  scan_id=scan_id, make_figures=False)
            except FileNotFoundError:
                continue
            summary[obj] = {k: v for k, v in res.items() if k.endswith("_mean")}
            coach.log(f"validation step {step} {obj}: " + "  ".join(f"{k} {v:.4f}" for k, v in summary[obj].items()))
        if summary:
            with open(Path(cfg.log.exp_dir) / f"{stem}_metrics.json", "w") as f:
                json.dump(summary, f, indent=1)
