"""End-to-end on the GPU through the reference-shaped surface: config -> Coach -> train -> checkpoints that
load back (tiny SD shape family, synthetic image folder, gradient accumulation 2)."""
import math

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def test_coach_mode0_trains_and_saves(tmp_path):
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.checkpoint_handler import CheckpointHandler
    from view_neti_amd.compat.coach import Coach
    root = tmp_path / "toys"
    root.mkdir()
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(root / f"{i}.png")
    cfg = C.parse(C.RunConfig, [
        "--data.train_data_dir", str(root), "--data.placeholder_object_token", "<toy>", "--data.resolution", "64",
        "--data.dataloader_num_workers", "0", "--model.word_embedding_dim", "128", "--model.arch_view_net", "15",
        "--model.arch_view_disable_tl", "False", "--model.arch_mlp_hidden_dims", "64", "--model.use_nested_dropout",
        "False", "--optim.max_train_steps", "4", "--optim.train_batch_size", "2",
        "--optim.gradient_accumulation_steps", "2", "--optim.mixed_precision", "fp16", "--log.save_steps", "2",
        "--eval.validation_steps", "2", "--eval.num_denoising_steps", "2", "--eval.num_validation_images", "2",
        "--eval.validation_seeds", "[0, 1]", "--log.exp_dir", str(tmp_path / "out"), "--log.exp_name", "run"])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    p0 = coach.engine.params.clone()
    coach.train()
    out = cfg.log.exp_dir
    assert (out / "config.yaml").exists() and (out / "logs" / "log.txt").exists()
    for name in ("learned_embeds-steps-2.bin", "mapper-steps-2_object.pt", "learned_embeds-final.bin",
                 "mapper-final_object.pt"):
        assert (out / name).exists(), name
    assert coach.engine.opt_step.item() == 4 and not torch.equal(p0, coach.engine.params)
    # validation grids at steps 2 and 4 (validate.py:296-309): 4 prompt templates x 2 seeds, generated with the live
    # mapper parameters (the validator's engine aliases the trainer's bucket)
    assert coach.validator.engine.text.mo.params.data_ptr() == coach.engine.params.data_ptr()
    for st in (2, 4):
        for i in range(4):
            g = Image.open(out / f"validation-iter_{st}-denoisesteps_2_upsample_1_imgs_t2i_{i}.png")
            assert g.size == (2 * 64, 64)
    # lr rule of coach.py:728-733: 1e-3 * accum(2) * bs(2) * world(1)
    assert abs(float(coach.engine.hyper[0]) - 4e-3) < 1e-9
    tok_id = coach.placeholder_object_token_ids[0]
    cfg2, lookup = CheckpointHandler.load_mapper(out / "mapper-final_object.pt", "object", ["<toy>"], [tok_id])
    from view_neti_amd.engine.text import flatten_mapper_state
    flat = flatten_mapper_state(lookup[tok_id].mapper_state())
    assert torch.allclose(flat, coach.engine.params.cpu(), atol=0, rtol=0)
    assert abs(cfg2.model.target_norm_object - cfg.model.target_norm_object) < 1e-6
    emb = torch.load(out / "learned_embeds-final.bin")
    assert list(emb) == ["<toy>"] and emb["<toy>"].shape == (128,)


M3_YAML = """
learnable_mode: 3
log: {{exp_name: m3, exp_dir: {out}, save_steps: 3}}
data: {{train_data_dir: data/dtu/Rectified, train_data_subsets: [scan65, scan125], super_category_object_tokens: [object, object],
       placeholder_object_tokens: [<skull>, <statue>], placeholder_object_token: <object>, dataloader_num_workers: 0,
       camera_representation: dtu-12d, dtu_subset: 3, dtu_lighting: 3, dtu_preprocess_key: 0, augmentation_key: 0,
       resolution: 64}}
model: {{arch_mlp_hidden_dims: 128, use_nested_dropout: True, nested_dropout_prob: 0.5, word_embedding_dim: 128,
        arch_view_net: 15, arch_view_disable_tl: False, pe_sigma_exp_key: 2, output_bypass_alpha_view: 5,
        output_bypass_alpha_object: 5, bypass_unconstrained_view: True}}
eval: {{validation_seeds: [0, 1], num_validation_images: 2, eval_placeholder_object_tokens: [<skull>]}}
optim: {{max_train_steps: 6, train_batch_size: 2, gradient_accumulation_steps: 1, mixed_precision: fp16}}
"""


def test_coach_mode3_multi_scene(tmp_path, monkeypatch):
    """BASELINE config 4 in miniature: two DTU scenes -> two object mappers (hidden 128) + one view mapper,
    nested dropout on, unconstrained view bypass; every batch is single-scene, the scene changes between
    optimizer steps, checkpoints hold every mapper."""
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.checkpoint_handler import CheckpointHandler
    from view_neti_amd.compat.coach import Coach
    from view_neti_amd.compat.dataset import TextualInversionDataset
    from view_neti_amd.engine.text import flatten_mapper_state
    monkeypatch.chdir(tmp_path)
    cal = tmp_path / "data" / "dtu" / "Calibration" / "cal18"
    cal.mkdir(parents=True)
    rng = np.random.RandomState(1)
    mats = rng.randn(49, 3, 4) * np.array([[1e3, 1e3, 1e3, 1e5]])
    for i in range(49):
        np.savetxt(cal / f"pos_{i + 1:03d}.txt", mats[i])
    names = [TextualInversionDataset.dtu_cam_and_lighting_to_fname(c, "3") for c in range(49)]
    for scan in ("scan65", "scan125"):
        d = tmp_path / "data" / "dtu" / "Rectified" / scan
        d.mkdir(parents=True)
        for n in names:
            # DTU frames are 1600x1200 (key 0 pads to 1600x1600, then resizes to 512x512)
            Image.fromarray(rng.randint(0, 255, (1200, 1600, 3), dtype=np.uint8)).save(d / n)
    y = tmp_path / "m3.yaml"
    y.write_text(M3_YAML.format(out=str(tmp_path / "out")))
    cfg = C.parse(C.RunConfig, ["--config_path", str(y)])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    eng = coach.engine
    assert eng.n_objects == 2 and eng.text.mo.hidden == 128 and eng.text.mo.nested_dropout_prob == 0.5
    assert eng.text.mv.unconstrained and not eng.text.mo.unconstrained
    scenes = []
    orig = eng.set_batch
    def spy(*a, **k):
        scenes.append(k["object_index"])
        return orig(*a, **k)
    eng.set_batch = spy
    p0 = eng.params.clone()
    coach.train()
    assert eng.opt_step.item() == 6 and len(scenes) == 6
    assert set(scenes) == {0, 1}, f"np.random scene sampler never switched scene in 6 steps: {scenes}"
    n = eng.n_obj
    for k in (0, 1):
        assert not torch.equal(p0[k * n:(k + 1) * n], eng.params[k * n:(k + 1) * n])
    assert torch.isfinite(eng.params).all()
    out = cfg.log.exp_dir
    for name in ("mapper-steps-3_object.pt", "mapper-steps-3_view.pt", "mapper-final_object.pt", "mapper-final_view.pt",
                 "learned_embeds-final.bin"):
        assert (out / name).exists(), name
    ids = coach.placeholder_object_token_ids
    _, lookup = CheckpointHandler.load_mapper(out / "mapper-final_object.pt", "object", ["<skull>", "<statue>"], ids)
    for k, tid in enumerate(ids):
        flat = flatten_mapper_state(lookup[tid].mapper_state())
        assert torch.equal(flat, eng.object_params(k).cpu()) and lookup[tid].hidden == 128
    _, view = CheckpointHandler.load_mapper(out / "mapper-final_view.pt", "view")
    assert torch.equal(flatten_mapper_state(view.mapper_state()), eng.view_params_flat().cpu())
    assert view.bypass_unconstrained and view.use_nested_dropout


def test_train_then_generate(tmp_path, monkeypatch):
    """the loop a user runs: train (mode 2: object + view mapper on a DTU-shaped scene) -> checkpoints ->
    `load_inference` rebuilds tokenizer/mappers from the run directory -> `sd_pipeline_call` generates."""
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.coach import Coach
    from view_neti_amd.compat.dataset import TextualInversionDataset
    from view_neti_amd.compat.inference import load_inference
    from view_neti_amd.compat.sd_pipeline_call import sd_pipeline_call
    from view_neti_amd.engine.text import flatten_mapper_state
    monkeypatch.chdir(tmp_path)
    cal = tmp_path / "data" / "dtu" / "Calibration" / "cal18"
    cal.mkdir(parents=True)
    rng = np.random.RandomState(1)
    mats = rng.randn(49, 3, 4) * np.array([[1e3, 1e3, 1e3, 1e5]])
    for i in range(49):
        np.savetxt(cal / f"pos_{i + 1:03d}.txt", mats[i])
    scan = tmp_path / "data" / "dtu" / "Rectified" / "scan114"
    scan.mkdir(parents=True)
    for c in range(49):
        Image.fromarray(rng.randint(0, 255, (120, 160, 3), dtype=np.uint8)).save(
            scan / TextualInversionDataset.dtu_cam_and_lighting_to_fname(c, "3"))
    cfg = C.parse(C.RunConfig, [
        "--learnable_mode", "2", "--data.train_data_dir", str(scan), "--data.placeholder_object_token", "<object>",
        "--data.camera_representation", "dtu-12d", "--data.dtu_subset", "3", "--data.dtu_preprocess_key", "1",
        "--data.augmentation_key", "5", "--data.dataloader_num_workers", "0", "--model.word_embedding_dim", "128",
        "--model.arch_view_net", "15", "--model.arch_view_disable_tl", "False", "--model.arch_mlp_hidden_dims", "64",
        "--model.use_nested_dropout", "False", "--model.pe_sigma_exp_key", "2", "--optim.max_train_steps", "3",
        "--optim.train_batch_size", "1", "--optim.gradient_accumulation_steps", "1", "--optim.mixed_precision", "fp16",
        "--log.save_steps", "100", "--eval.validation_steps", "3", "--eval.num_denoising_steps", "2",
        "--eval.num_validation_images", "1", "--eval.validation_seeds", "[0]", "--log.exp_dir", str(tmp_path / "out"),
        "--log.exp_name", "m2"])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    coach.train()
    eng = coach.engine
    view_tok = coach.train_dataset.placeholder_view_tokens[0]
    # view-mode validation: camidx -> [image per seed] for the 3 training views + one grid per seed
    val = torch.load(cfg.log.exp_dir / "validation-iter_3-denoisesteps_2_numseeds_1_upsample_1.pt", weights_only=False)
    assert sorted(val) == sorted(coach.train_dataset.lookup_view_token_to_camidx[t]
                                 for t in coach.train_dataset.placeholder_view_tokens)
    assert all(len(v) == 1 and v[0].shape == (384, 512, 3) and v[0].dtype == np.uint8 for v in val.values())
    assert Image.open(cfg.log.exp_dir / "validation-iter_3-denoisesteps_2_numseeds_1_upsample_1_seed_0.png").size == (3 * 512, 384)
    pipe, pm = load_inference(cfg.log.exp_dir, "mapper-final", batch=1)
    ie = pipe.engine
    assert (ie.h, ie.w) == (48, 64)  # dtu_preprocess_key 1 -> 384 x 512
    assert torch.equal(ie.text.mo.params.cpu(), eng.object_params(0).cpu())
    assert torch.equal(ie.text.mv.params.cpu(), eng.view_params_flat().cpu())
    # all 49 DTU views are addressable at inference (novel views: neti_mapper.py:440-468), so the ids differ
    # from training; what matters is that each id maps to its token string and the right mapper
    assert len(pm.placeholder_view_token_ids) == 49
    assert pm.placeholder_object_token_ids == [pipe.tokenizer.convert_tokens_to_ids("<object>")]
    emb = pm.embed_prompt(f"{view_tok}. A photo of a <object>")
    assert int(emb.input_ids_placeholder_view) == pipe.tokenizer.convert_tokens_to_ids(view_tok)
    # the camera parameters the view mapper sees at inference == those the Coach fed it in training
    want = coach._view_params(torch.tensor([coach.tokenizer.convert_tokens_to_ids(view_tok)]))
    assert torch.allclose(emb.view_params, want, atol=1e-6)
    out = sd_pipeline_call(pipe, emb, num_inference_steps=4, guidance_scale=3.0,
                           generator=torch.Generator().manual_seed(0))
    im = out.images[0]
    assert im.size == (512, 384)
    a = np.asarray(im)
    out2 = sd_pipeline_call(pipe, emb, num_inference_steps=4, guidance_scale=3.0,
                            generator=torch.Generator().manual_seed(0), output_type="np", return_dict=False)[0]
    assert out2.shape == (1, 384, 512, 3) and np.isfinite(out2).all() and out2.min() >= 0 and out2.max() <= 1
    assert np.abs(a.astype(np.int32) - (out2[0] * 255).round().astype(np.int32)).mean() < 3.0  # same seed, same image
    # ---- learnable_mode 5 (BASELINE config 5's training side): a NEW object mapper is trained against the view mapper
    #      learned above, which stays frozen (coach.py:554-592,745-747) and keeps the bypass alpha it was trained with ----
    view_ckpt = cfg.log.exp_dir / "mapper-final_view.pt"
    cfg5 = C.parse(C.RunConfig, [
        "--learnable_mode", "5", "--data.train_data_dir", str(scan), "--data.placeholder_object_token", "<object>",
        "--data.camera_representation", "dtu-12d", "--data.dtu_subset", "1", "--data.dtu_preprocess_key", "1",
        "--data.dataloader_num_workers", "0", "--model.word_embedding_dim", "128", "--model.arch_view_net", "15",
        "--model.arch_view_disable_tl", "False", "--model.arch_mlp_hidden_dims", "64", "--model.use_nested_dropout", "False",
        "--model.pe_sigma_exp_key", "2", "--model.pretrained_view_mapper", str(view_ckpt), "--model.output_bypass_alpha_view",
        "0.7", "--optim.max_train_steps", "2", "--optim.train_batch_size", "1", "--optim.gradient_accumulation_steps", "1",
        "--optim.mixed_precision", "fp16", "--log.save_steps", "100", "--eval.validation_steps", "100",
        "--log.exp_dir", str(tmp_path / "out"), "--log.exp_name", "m5"])
    cfg5.log.exp_dir = cfg5.log.exp_dir / cfg5.log.exp_name
    cfg5.log.logging_dir = cfg5.log.exp_dir / cfg5.log.logging_dir
    torch.manual_seed(cfg5.seed)
    coach5 = Coach(cfg5)
    e5 = coach5.engine
    assert e5.view_params_flat().numel() == 0, "the frozen view mapper is not in the trainable bucket"
    assert torch.equal(e5.text.mv.params.cpu(), eng.view_params_flat().cpu()), "pretrained view mapper loaded"
    # quirk Q8: the loaded mapper carries the (object) alpha of ITS training config (0.2), not this run's 0.7
    assert abs(e5.text.mv.alpha - 0.2) < 1e-9
    pv0, po0 = e5.text.mv.params.clone(), e5.params.clone()
    coach5.train()
    assert torch.equal(e5.text.mv.params, pv0) and not torch.equal(e5.params, po0) and e5.opt_step.item() == 2
    assert (cfg5.log.exp_dir / "mapper-final_object.pt").exists()


@pytest.mark.parametrize("aug_key,flip", [(7, True), (5, False)])
def test_coach_device_input_pipeline_matches_host(tmp_path, aug_key, flip):
    """cfg.data.device_input_pipeline: the same seeded run feeds the step the same pixels whether the
    resize / flip / augmentation pipeline runs in PIL on the host or as HIP kernels on cached images, and trains."""
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.coach import Coach
    root = tmp_path / "toys"
    root.mkdir()
    rng = np.random.RandomState(1)
    for i in range(3):
        a = rng.randint(0, 255, (12, 16, 3), dtype=np.uint8)
        Image.fromarray(a).resize((160, 120), Image.BICUBIC).save(root / f"{i}.png")

    def make(device_pipe, name):
        cfg = C.parse(C.RunConfig, [
            "--data.train_data_dir", str(root), "--data.placeholder_object_token", "<toy>", "--data.resolution", "64",
            "--data.dataloader_num_workers", "0", "--data.augmentation_key", str(aug_key),
            "--data.device_input_pipeline", str(device_pipe), "--model.word_embedding_dim", "128",
            "--model.arch_view_net", "15", "--model.arch_view_disable_tl", "False", "--model.arch_mlp_hidden_dims", "64",
            "--model.use_nested_dropout", "False", "--optim.max_train_steps", "3", "--optim.train_batch_size", "2",
            "--optim.mixed_precision", "fp16", "--log.save_steps", "100", "--log.exp_dir", str(tmp_path / name),
            "--log.exp_name", "run"])
        cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
        cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
        torch.manual_seed(cfg.seed)
        c = Coach(cfg)
        c.train_dataset.flip_p = 0.5 if flip else 0.0  # Coach never forwards flip_p (reference quirk Q10)
        return c

    host, dev = make(False, "host"), make(True, "dev")
    assert dev.device_pipe is not None and host.device_pipe is None
    worst = 0.0
    for it in range(3):
        import random
        torch.manual_seed(1000 + it)
        random.seed(it)  # the caption template is drawn with python's `random` (dataset.py:630)
        bh = next(iter(host.train_dataloader))
        torch.manual_seed(1000 + it)
        random.seed(it)
        bd = next(iter(dev.train_dataloader))
        assert "pixel_values" in bh and "aug" in bd and "pixel_values" not in bd
        assert torch.equal(bh["input_ids"], bd["input_ids"]) and torch.equal(bh["image_idx"], bd["image_idx"])
        assert dev._pixels(bd) is None
        torch.cuda.synchronize()
        d = (dev.engine.pixel_values.cpu() - host._pixels(bh)).abs()
        assert d.max().item() <= 2 / 127.5 + 1e-6, f"iteration {it}: max pixel difference {d.max().item() * 127.5:.2f} LSB"
        worst = max(worst, (d > 1e-6).float().mean().item())
    assert worst < 0.02, worst
    p0 = dev.engine.params.clone()
    dev.train()
    assert dev.engine.opt_step.item() == 3 and not torch.equal(p0, dev.engine.params)
    assert np.isfinite(dev.engine.loss())


def test_coach_reference_default_mapper_config_trains(tmp_path):
    """SURVEY a5': `Coach(RunConfig())`-style defaults for the mapper — arch_view_net 0 / arch_view_disable_tl True /
    arch_mlp_hidden_dims 128 / nested dropout on (training/config.py:89-130) — i.e. the LEGACY object mapper with its
    trainable input_layer: trains, checkpoints, and the checkpoint (pickled NeTIPositionalEncoding included) loads back
    into the same parameters."""
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.checkpoint_handler import CheckpointHandler
    from view_neti_amd.compat.coach import Coach
    from view_neti_amd.engine.text import flatten_mapper_state
    root = tmp_path / "toys"
    root.mkdir()
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(root / f"{i}.png")
    cfg = C.parse(C.RunConfig, [
        "--data.train_data_dir", str(root), "--data.placeholder_object_token", "<toy>", "--data.resolution", "64",
        "--data.dataloader_num_workers", "0", "--model.word_embedding_dim", "128", "--optim.max_train_steps", "3",
        "--optim.train_batch_size", "2", "--optim.gradient_accumulation_steps", "1", "--optim.mixed_precision", "fp16",
        "--log.save_steps", "100", "--eval.validation_steps", "100", "--log.exp_dir", str(tmp_path / "out"),
        "--log.exp_name", "legacy"])
    assert cfg.model.arch_view_net == 0 and cfg.model.arch_view_disable_tl and cfg.model.use_nested_dropout
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    eng = coach.engine
    n_std = 160 * 128 + 128 * 3 + 128 * 128 + 128 * 3 + 256 * 128 + 256
    assert eng.params.numel() == n_std + 160 * 2048 + 160 and eng.text.mo.legacy_w_pe.shape == (1024, 2)
    p0 = eng.params.clone()
    coach.train()
    assert eng.opt_step.item() == 3 and torch.isfinite(eng.params).all()
    moved_in = (eng.params[n_std:] != p0[n_std:]).float().mean().item()
    assert moved_in > 0.9, "the input_layer is trained too"
    tok_id = coach.placeholder_object_token_ids[0]
    _, lookup = CheckpointHandler.load_mapper(cfg.log.exp_dir / "mapper-final_object.pt", "object", ["<toy>"], [tok_id])
    m = lookup[tok_id]
    assert m.legacy and torch.equal(m.encoder.w, coach.mapper_object_lookup[tok_id].encoder.w)
    assert torch.equal(flatten_mapper_state(m.mapper_state()), eng.params.cpu())


M1_YAML = """
learnable_mode: 1
log: {{exp_name: m1, exp_dir: {out}, save_steps: 3}}
data: {{train_data_dir: data/dtu/Rectified/scan65, placeholder_object_token: <object>, fixed_object_token_or_path: statue,
       dataloader_num_workers: 0, camera_representation: dtu-12d, dtu_subset: 3, dtu_lighting: 3, dtu_preprocess_key: 0,
       augmentation_key: 0, resolution: 64}}
model: {{arch_mlp_hidden_dims: 64, use_nested_dropout: False, word_embedding_dim: 128, arch_view_net: 15,
        arch_view_disable_tl: False, pe_sigma_exp_key: 2, output_bypass_alpha_view: 5}}
eval: {{validation_steps: 3, num_denoising_steps: 2, num_validation_images: 1, validation_seeds: [0]}}
optim: {{max_train_steps: 4, train_batch_size: 2, gradient_accumulation_steps: 1, mixed_precision: fp16}}
"""


def test_coach_mode1_view_mapper_only(tmp_path, monkeypatch):
    """learnable_mode 1 (training/coach.py:493,553-584; dataset.py:654-668): only the view mapper trains, the object is a
    vocabulary word — captions "<view_x>. A photo of a statue", no object placeholder in the batch, no object checkpoint."""
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.checkpoint_handler import CheckpointHandler
    from view_neti_amd.compat.coach import Coach
    from view_neti_amd.compat.dataset import TextualInversionDataset
    from view_neti_amd.engine.text import flatten_mapper_state
    monkeypatch.chdir(tmp_path)
    cal = tmp_path / "data" / "dtu" / "Calibration" / "cal18"
    cal.mkdir(parents=True)
    rng = np.random.RandomState(2)
    mats = rng.randn(49, 3, 4) * np.array([[1e3, 1e3, 1e3, 1e5]])
    for i in range(49):
        np.savetxt(cal / f"pos_{i + 1:03d}.txt", mats[i])
    d = tmp_path / "data" / "dtu" / "Rectified" / "scan65"
    d.mkdir(parents=True)
    for c in range(49):
        Image.fromarray(rng.randint(0, 255, (1200, 1600, 3), dtype=np.uint8)).save(
            d / TextualInversionDataset.dtu_cam_and_lighting_to_fname(c, "3"))
    y = tmp_path / "m1.yaml"
    y.write_text(M1_YAML.format(out=str(tmp_path / "out")))
    cfg = C.parse(C.RunConfig, ["--config_path", str(y)])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    eng = coach.engine
    assert coach.mapper_object_lookup is None and coach.mapper_view is not None
    ex = coach.train_dataset[0]
    assert int(ex["input_ids_placeholder_object"]) == -1 and ex["text"].endswith(". A photo of a statue")
    v0 = eng.view_params_flat().clone()
    coach.train()
    assert eng.opt_step.item() == 4 and torch.isfinite(eng.params).all()
    assert not torch.equal(v0, eng.view_params_flat()), "the view mapper must train"
    # the stand-in object mapper is inert: no prompt reaches it, so its gradient segment is exactly zero (no BOS-row dX)
    assert float(eng.grads[: eng.n_all_obj].abs().max()) == 0.0
    out = cfg.log.exp_dir
    # validate.py:455: mode 1 validates with the view-token prompts and the vocabulary word (camidx -> images dict + grid)
    val = torch.load(out / "validation-iter_3-denoisesteps_2_numseeds_1_upsample_1.pt", weights_only=False)
    ds = coach.train_dataset
    assert sorted(val) == sorted(ds.lookup_view_token_to_camidx[t] for t in ds.placeholder_view_tokens)  # the training views
    assert tuple(val[min(val)][0].shape) == (*coach._image_hw(), 3)
    assert (out / "mapper-final_view.pt").exists() and not (out / "mapper-final_object.pt").exists()
    _, view = CheckpointHandler.load_mapper(out / "mapper-final_view.pt", "view")
    assert torch.equal(flatten_mapper_state(view.mapper_state()), eng.view_params_flat().cpu())


def test_coach_vae_moment_cache(tmp_path):
    """`--data.cache_vae_moments True` (extension): mode 0, augmentation_key 0 — every image is encoded once, later batches
    replay the cached-moments graph; refused where the dataset is not deterministic."""
    from view_neti_amd.compat import config as C
    from view_neti_amd.compat.coach import Coach
    root = tmp_path / "toys"
    root.mkdir()
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (90, 120, 3), dtype=np.uint8)).save(root / f"{i}.png")
    base = ["--data.train_data_dir", str(root), "--data.placeholder_object_token", "<toy>", "--data.resolution", "64",
            "--data.dataloader_num_workers", "0", "--model.word_embedding_dim", "128", "--model.arch_view_net", "15",
            "--model.arch_view_disable_tl", "False", "--model.arch_mlp_hidden_dims", "64", "--model.use_nested_dropout", "False",
            "--optim.max_train_steps", "6", "--optim.train_batch_size", "2", "--optim.gradient_accumulation_steps", "1",
            "--optim.mixed_precision", "fp16", "--log.save_steps", "100", "--eval.validation_steps", "100",
            "--data.cache_vae_moments", "True", "--log.exp_name", "run"]
    cfg = C.parse(C.RunConfig, base + ["--log.exp_dir", str(tmp_path / "out")])
    cfg.log.exp_dir = cfg.log.exp_dir / cfg.log.exp_name
    cfg.log.logging_dir = cfg.log.exp_dir / cfg.log.logging_dir
    torch.manual_seed(cfg.seed)
    coach = Coach(cfg)
    eng = coach.engine
    assert eng.n_cache == 3
    runs = [0]
    fwd = eng.vae.forward
    eng.vae.forward = lambda: (runs.__setitem__(0, runs[0] + 1), fwd())[1]
    coach.train()
    assert sorted(eng._cached_images) == [0, 1, 2] and eng.graph_a_c is not None
    assert int(eng.opt_step.item()) == 6 and math.isfinite(eng.loss())
    assert runs[0] <= 3, "the encoder ran eagerly more often than the capture's warm-up + trace passes allow"
    cfg2 = C.parse(C.RunConfig, base + ["--log.exp_dir", str(tmp_path / "out2"), "--data.augmentation_key", "5"])
    cfg2.log.exp_dir = cfg2.log.exp_dir / cfg2.log.exp_name
    cfg2.log.logging_dir = cfg2.log.exp_dir / cfg2.log.logging_dir
    with pytest.raises(ValueError, match="deterministic"):
        Coach(cfg2)
