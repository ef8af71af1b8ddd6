"""CPU tests of the reference-compatible surface: config (+ mini-pyrallis), tokenizer stand-in, dataset
sample contract, mapper modules and the checkpoint format."""
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

from view_neti_amd.compat import config as C
from view_neti_amd.compat.checkpoint_handler import CheckpointHandler
from view_neti_amd.compat.constants import IMAGENET_TEMPLATES_SMALL, UNET_LAYERS
from view_neti_amd.compat.dataset import TextualInversionDataset
from view_neti_amd.compat.neti_modules import FourierPositionalEncodingNDims, NeTIMapper
from view_neti_amd.compat.tokenizer import HashTokenizer

YAML = """
learnable_mode: 2
log: {exp_name: t, exp_dir: results, save_steps: 1500}
data: {train_data_dir: data/x, placeholder_object_token: <object>, dataloader_num_workers: 0,
       camera_representation: dtu-12d, dtu_subset: 0}
model: {arch_mlp_hidden_dims: 64, use_nested_dropout: False, word_embedding_dim: 1024, arch_view_net: 15,
        arch_view_disable_tl: False, pe_sigma_exp_key: 2, output_bypass_alpha_view: 5, output_bypass_alpha_object: 5,
        pe_sigmas: {sigma_t: 0.5, sigma_l: 9.0, sigma_phi: 2.0, sigma_theta: 2.0}}
eval: {validation_seeds: [0, 1], num_validation_images: 2}
optim: {max_train_steps: 10, train_batch_size: 3, gradient_accumulation_steps: 3}
"""


def test_config_defaults_and_post_init(tmp_path):
    d = C.RunConfig(data=C.DataConfig(train_data_dir="x"))
    assert d.optim.train_batch_size == 3 and d.optim.gradient_accumulation_steps == 3 and d.optim.mixed_precision == "no"
    assert d.model.arch_view_net == 0 and d.model.arch_view_disable_tl is True and d.model.word_embedding_dim == 768
    assert isinstance(d.model.pe_sigmas, C.PESigmas) and d.model.pe_sigmas.sigma_dtu12 == 2.0
    p = tmp_path / "c.yaml"
    p.write_text(YAML)
    cfg = C.parse(C.RunConfig, ["--config_path", str(p), "--optim.learning_rate", "5e-4", "--log.overwrite_ok"])
    s = cfg.model.pe_sigmas
    # App. C Q3: sigma_t / sigma_l come from the exp keys (YAML values ignored), phi copied, exp_key 2 -> 0.5
    assert (s.sigma_t, s.sigma_l, s.sigma_theta, s.sigma_phi, s.sigma_r, s.sigma_dtu12) == (0.03, 2.0, 2.0, 2.0, 2.0, 0.5)
    assert cfg.optim.learning_rate == 5e-4 and cfg.log.overwrite_ok is True and cfg.log.exp_dir.name == "results"
    enc = C.encode(cfg)
    assert isinstance(enc["log"]["exp_dir"], str) and enc["model"]["pe_sigmas"]["sigma_dtu12"] == 0.5
    assert C.encode(C.decode(C.RunConfig, enc)) == enc  # decode(encode(.)) is the identity on parsed configs
    with pytest.raises(AssertionError):
        C.EvalConfig(validation_seeds=[0], num_validation_images=2)
    with pytest.warns(UserWarning):
        C.RunConfig(optim=C.OptimConfig(train_batch_size=4))


def test_constants():
    assert len(UNET_LAYERS) == 16 and UNET_LAYERS[6] == "MID" and UNET_LAYERS[-1] == "OUT11"
    assert len(IMAGENET_TEMPLATES_SMALL) == 27 and all("{}" in t for t in IMAGENET_TEMPLATES_SMALL)


def test_tokenizer_contract():
    tk = HashTokenizer()
    assert tk.add_tokens(["<view_dtu12d_cam3_1p5_2>", "<obj>"]) == 2 and tk.add_tokens(["<obj>"]) == 0
    ids = tk("<view_dtu12d_cam3_1p5_2>. A photo of a <obj>").input_ids
    assert ids.shape == (1, 77) and ids[0, 0] == 49406 and ids[0, -1] == 49407
    assert (ids[0] == 49408).sum() == 1 and (ids[0] == 49409).sum() == 1 and len(tk) == 49410
    assert len(tk.encode("object", add_special_tokens=False)) == 1


def _make_images(root, names, size=(100, 80)):
    root.mkdir(parents=True, exist_ok=True)
    rng = np.random.RandomState(0)
    for n in names:
        Image.fromarray(rng.randint(0, 255, (size[1], size[0], 3), dtype=np.uint8)).save(root / n)


def test_dataset_mode0(tmp_path):
    _make_images(tmp_path / "toys", ["a.png", "b.jpg", "c.txt.png"])
    tk = HashTokenizer()
    tk.add_tokens(["<toy>"])
    ds = TextualInversionDataset(tmp_path / "toys", tk, learnable_mode=0, size=64, repeats=5, placeholder_object_token="<toy>")
    assert len(ds) == 15
    ex = ds[4]
    assert ex["pixel_values"].shape == (3, 64, 64) and ex["pixel_values"].dtype == torch.float32
    assert -1.0 <= ex["pixel_values"].min() and ex["pixel_values"].max() <= 1.0
    assert ex["input_ids"].shape == (77,) and int(ex["input_ids_placeholder_view"]) == -1
    assert (ex["input_ids"] == ex["input_ids_placeholder_object"]).sum() == 1 and "<toy>" in ex["text"]
    with pytest.raises(ValueError):
        TextualInversionDataset(tmp_path / "toys", tk, augmentation_key=9)


def test_augmentation_pipelines(tmp_path):
    """dataset.py:238-316 restated without torchvision: every key keeps the frame size and value range, is
    reproducible from the torch seed, and each primitive does what its torchvision namesake does."""
    from view_neti_amd.compat import augment as A
    _make_images(tmp_path / "toys", ["a.png"], size=(120, 90))
    tk = HashTokenizer()
    tk.add_tokens(["<toy>"])
    base = TextualInversionDataset(tmp_path / "toys", tk, learnable_mode=0, size=64, placeholder_object_token="<toy>")[0]
    for key in range(1, 9):
        ds = TextualInversionDataset(tmp_path / "toys", tk, learnable_mode=0, size=64, placeholder_object_token="<toy>",
                                     augmentation_key=key)
        torch.manual_seed(3)
        a = ds[0]["pixel_values"]
        torch.manual_seed(3)
        b = ds[0]["pixel_values"]
        assert a.shape == (3, 64, 64) and torch.equal(a, b) and -1.0 <= a.min() and a.max() <= 1.0
        diffs = []
        for seed in range(6):
            torch.manual_seed(seed)
            diffs.append((ds[0]["pixel_values"] - base["pixel_values"]).abs().mean().item())
        assert max(diffs) > 0, f"augmentation_key {key} never changed the image"
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 255, (48, 64, 3), dtype=np.uint8))
    # hue shift by a whole turn is the identity up to HSV quantisation; +0 is exactly RGB->HSV->RGB
    assert np.abs(np.asarray(A.adjust_hue(img, 0.0), dtype=int) - np.asarray(img.convert("HSV").convert("RGB"), dtype=int)).max() == 0
    g = np.asarray(A.to_grayscale3(img))
    assert (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()
    # a constant image is a fixed point of the blur (reflect padding, normalised kernel)
    const = Image.fromarray(np.full((20, 30, 3), 77, dtype=np.uint8))
    assert (np.asarray(A.gaussian_blur(const)) == 77).all()
    # sigma in [0.1, 0.2] makes the 5-tap kernel almost a delta: the blur moves pixels by < 1 grey level
    assert np.abs(np.asarray(A.gaussian_blur(img), dtype=int) - np.asarray(img, dtype=int)).max() <= 1
    filled = 0
    for seed in range(5):
        torch.manual_seed(seed)
        r = np.asarray(A.random_rotation(img, 10.0, fill=1))
        assert r.shape == (48, 64, 3)
        filled += sum(int((r[y, x] == 1).all()) for y in (0, -1) for x in (0, -1))
    assert filled > 0  # rotated-in corners carry the constant fill value 1
    torch.manual_seed(0)
    c = A.random_resized_crop(img, (24, 32), (0.7, 1.3))
    assert c.size == (32, 24)
    # scale > 1 can only succeed through the fallback (whole image): still the right output size
    assert A.random_resized_crop(img, (24, 32), (1.5, 1.6)).size == (32, 24)


def test_dataset_dtu_view_mode(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cal = tmp_path / "data" / "dtu" / "Calibration" / "cal18"
    cal.mkdir(parents=True)
    rng = np.random.RandomState(1)
    mats = rng.randn(49, 3, 4) * np.array([[1e3, 1e3, 1e3, 1e5]])
    for i in range(49):
        np.savetxt(cal / f"pos_{i + 1:03d}.txt", mats[i])
    scan = tmp_path / "data" / "dtu" / "Rectified" / "scan114"
    names = [TextualInversionDataset.dtu_cam_and_lighting_to_fname(c, "3") for c in range(49)]
    _make_images(scan, names + [TextualInversionDataset.dtu_cam_and_lighting_to_fname(0, "1")], size=(160, 120))
    tk = HashTokenizer()
    ds = TextualInversionDataset(scan, tk, camera_representation="dtu-12d", learnable_mode=2, dtu_subset=3,
                                 dtu_lighting=3, dtu_preprocess_key=1, placeholder_object_token="<object>")
    assert ds.num_images == 3 and len(ds.placeholder_view_tokens) == 3  # DTU_TRAIN_IDX[:3] = 25, 22, 28
    tk.add_tokens(ds.placeholder_tokens)
    ex = ds[0]
    assert ex["pixel_values"].shape == (3, 384, 512)
    assert ex["text"].startswith("<view_dtu12d_cam22_") and ex["text"].endswith(". A photo of a <object>")
    # token <-> params round trip is the 4-decimal quantisation of the calibration matrix (App. C Q15)
    p, key = TextualInversionDataset.dtu_token_to_cam_params(ds.placeholder_view_tokens[0], cam_idx_as_int=True)
    assert key == 22 and np.allclose(p.numpy(), mats[22].flatten(), atol=6e-5 + 1e-7 * np.abs(mats[22]).max())
    assert TextualInversionDataset.dtu_get_train_idxs(-3) == list(range(12, 36, 3))


def test_mapper_module_and_checkpoint_format(tmp_path):
    torch.manual_seed(3)
    m1 = NeTIMapper("object", 768, 64, 0.4, placeholder_object_token="<toy>")
    m2 = NeTIMapper("object", 768, 64, 0.4, placeholder_object_token="<toy2>")
    # App. C Q1: every mapper starts identical because the encoder re-seeds the global RNG
    assert all(torch.equal(a, b) for a, b in zip(m1.mapper_state().values(), m2.mapper_state().values()))
    assert sum(v.numel() for v in m1.mapper_state().values()) == 108416 and "encoder.w" not in m1.state_dict()
    assert sum(v.numel() for v in NeTIMapper("object", 1024, 64).mapper_state().values()) == 141696
    w, b = m1(torch.tensor([10, 500]), torch.tensor([0, 7]))
    assert w.shape == (2, 768) and torch.allclose(w.norm(dim=-1), torch.full((2,), 0.4), atol=1e-5)
    cfg = C.RunConfig(data=C.DataConfig(train_data_dir="x", placeholder_object_token="<toy>"),
                      model=C.ModelConfig(arch_view_net=15, arch_view_disable_tl=False, arch_mlp_hidden_dims=64,
                                          target_norm_object=0.4, use_nested_dropout=False))
    h = CheckpointHandler(cfg, [], [], ["<toy>"], [49408], tmp_path)
    E = torch.randn(49409, 768)
    h.save_model(E, {49408: m1}, None, "learned_embeds-steps-5.bin", "mapper-steps-5.pt")
    emb = torch.load(tmp_path / "learned_embeds-steps-5.bin")
    assert list(emb) == ["<toy>"] and torch.equal(emb["<toy>"], E[49408])
    ck = torch.load(tmp_path / "mapper-steps-5_object.pt", weights_only=False)
    # the reference's layout + one extra top-level key its loader never reads (extension fields, synthetic marker)
    assert set(ck) == {"cfg", "mappers", "vneti_ext"} and list(ck["mappers"]) == [49408]
    assert ck["vneti_ext"]["synthetic_sd_weights"] is False
    assert "device_input_pipeline" not in ck["cfg"]["data"] and "allow_synthetic_weights" not in ck["cfg"]["model"]
    entry = ck["mappers"][49408]
    assert set(entry) == {"state_dict", "encoder", "placeholder_object_token"} and entry["placeholder_object_token"] == "<toy>"
    assert list(entry["state_dict"]) == ["net.0.weight", "net.0.bias", "net.1.weight", "net.1.bias", "net.3.weight",
                                         "net.3.bias", "net.4.weight", "net.4.bias", "output_layer.0.weight",
                                         "output_layer.0.bias"]
    assert type(entry["encoder"]).__module__ == "models.positional_encoding"
    assert type(entry["encoder"]).__name__ == "FourierPositionalEncodingNDims" and ck["cfg"]["model"]["arch_view_net"] == 15
    cfg2, lookup = CheckpointHandler.load_mapper(tmp_path / "mapper-steps-5_object.pt", "object", ["<toy>"], [49408])
    assert cfg2.model.target_norm_object == 0.4
    w2, _ = lookup[49408](torch.tensor([10, 500]), torch.tensor([0, 7]))
    assert torch.allclose(w, w2)


def test_lr_schedules():
    """compat/lr_schedule.py against the defining properties of diffusers' get_scheduler shapes (coach.py:759-770):
    warm-up ramps linearly from 0, constant stays 1, linear/cosine/polynomial end at ~0 at num_training_steps, the
    scheduler advances `world` ticks per optimizer step (accelerate's AcceleratedScheduler) with totals x accum."""
    import math
    from view_neti_amd.compat.lr_schedule import LRSchedule, lr_lambda
    assert all(lr_lambda("constant", s, 10, 100) == 1.0 for s in (0, 5, 100, 1000))
    assert lr_lambda("constant_with_warmup", 0, 10, 100) == 0.0 and lr_lambda("constant_with_warmup", 5, 10, 100) == 0.5
    assert lr_lambda("constant_with_warmup", 10, 10, 100) == 1.0
    assert lr_lambda("linear", 5, 10, 110) == 0.5 and lr_lambda("linear", 10, 10, 110) == 1.0
    assert abs(lr_lambda("linear", 60, 10, 110) - 0.5) < 1e-12 and lr_lambda("linear", 110, 10, 110) == 0.0
    assert abs(lr_lambda("cosine", 60, 10, 110) - 0.5) < 1e-12 and lr_lambda("cosine", 110, 10, 110) < 1e-12
    assert abs(lr_lambda("cosine", 35, 10, 110) - 0.5 * (1 + math.cos(math.pi * 0.25))) < 1e-12
    assert lr_lambda("cosine_with_restarts", 110, 10, 110) == 0.0 and lr_lambda("cosine_with_restarts", 10, 10, 110) == 1.0
    assert abs(lr_lambda("polynomial", 110, 10, 110, lr_init=1e-3) - 1e-7 / 1e-3) < 1e-12
    assert abs(lr_lambda("polynomial", 60, 10, 110, lr_init=1e-3) - (0.5 * (1e-3 - 1e-7) + 1e-7) / 1e-3) < 1e-12
    with pytest.raises(ValueError):
        lr_lambda("nope", 0, 0, 1)
    s = LRSchedule("linear", 4e-3, lr_warmup_steps=2, max_train_steps=10, grad_accum=3, world=2)
    assert s.warmup == 6 and s.total == 30 and not s.constant
    assert s.lr(0) == 0.0 and abs(s.lr(3) - 4e-3) < 1e-15          # 3 optimizer steps x 2 ranks = 6 ticks = end of warm-up
    assert abs(s.lr(9) - 4e-3 * (30 - 18) / 24) < 1e-15
    assert LRSchedule("constant", 1e-3, 0, 10, 1, 8).constant


def test_synthetic_weights_must_be_asked_for(monkeypatch):
    """a hub id such as the reference default cannot be resolved offline — that is an error, not a silent
    fall-back to random weights; the opt-in is explicit (config flag or environment)."""
    from view_neti_amd import sd_config as sc
    from view_neti_amd.compat import sd_weights
    monkeypatch.delenv("VNETI_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError, match="allow_synthetic_weights"):
        sd_weights.load_sd_weights(sc.tiny(), "CompVis/stable-diffusion-v1-4", device="cpu")
    with pytest.raises(FileNotFoundError):
        sd_weights.load_vae_decoder_weights(sc.tiny(), "CompVis/stable-diffusion-v1-4", device="cpu")
    assert sd_weights.synthetic_allowed(True)
    monkeypatch.setenv("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")
    assert sd_weights.synthetic_allowed(False)


def test_config_rejects_unknown_keys_and_keeps_ext_out_of_checkpoints():
    from view_neti_amd.compat import config as C
    with pytest.raises(ValueError, match="unknown configuration key"):
        C.parse(C.RunConfig, ["--optim.learning_rat", "1e-3"])
    cfg = C.parse(C.RunConfig, ["--data.device_input_pipeline", "true", "--model.allow_synthetic_weights", "true"])
    full, ref = C.encode(cfg), C.encode(cfg, include_ext=False)
    assert full["data"]["device_input_pipeline"] is True and "device_input_pipeline" not in ref["data"]
    assert "allow_synthetic_weights" not in ref["model"] and "placeholder_view_tokens" not in ref["data"]
    assert C.ext_fields(cfg) == {"data.device_input_pipeline": True, "data.cache_vae_moments": False,
                                 "model.allow_synthetic_weights": True}
    assert "cache_vae_moments" not in ref["data"]
    cfg.data.placeholder_view_tokens = ["<v>"]  # run-time attribute, never serialised (config.py:64 of the reference)
    assert "placeholder_view_tokens" not in C.encode(cfg)["data"]


# ------------------------------------------------------------------------------------------------
# artefacts produced by the REFERENCE's own classes load here (tests/golden/f2_*, made by
# oracle/make_golden.py::f2_reference_checkpoints in the checkpoint_handler.py:57-97 layout)
# ------------------------------------------------------------------------------------------------
_G = os.path.join(os.path.dirname(__file__), "golden")


def test_f2_reference_mapper_checkpoints_load_and_reproduce():
    exp = np.load(os.path.join(_G, "f2_expected.npz"))
    t, lay = torch.from_numpy(exp["t"]), torch.from_numpy(exp["l"])
    cfg, lookup = CheckpointHandler.load_mapper(os.path.join(_G, "f2_mapper-steps-7_object.pt"), "object", ["<obj>"], [90])
    assert list(lookup) == [90] and cfg.model.word_embedding_dim == 32 and cfg.learnable_mode == 2
    assert cfg.model.pe_sigmas.sigma_dtu12 == float(exp["sigma_dtu12"]) == 0.5   # pe_sigma_exp_key 2 survives the trip
    m = lookup[90]
    assert m.placeholder_object_token == "<obj>" and m.output_bypass_alpha == 5 and not m.training
    with torch.no_grad():
        w, b = m(t, lay)
    assert torch.allclose(w, torch.from_numpy(exp["word_obj"]), atol=1e-6)
    assert torch.allclose(b, torch.from_numpy(exp["bypass_obj"]), atol=1e-5)
    assert torch.equal(m.encoder.w, torch.from_numpy(exp["w_obj"]))  # regenerated from the seed == what the reference drew
    # the pickled encoder instance itself resolves through the `models.*` alias package and is usable
    raw = torch.load(os.path.join(_G, "f2_mapper-steps-7_object.pt"), map_location="cpu", weights_only=False)
    enc = raw["mappers"][90]["encoder"]
    assert type(enc).__module__ == "models.positional_encoding" and isinstance(enc, FourierPositionalEncodingNDims)
    x = torch.stack((t / 1000 * 2 - 1, lay / 16 * 2 - 1), 1)
    assert torch.allclose(enc(x), m.encoder(x), atol=1e-6)
    assert "encoder.w" not in raw["mappers"][90]["state_dict"]
    # view mapper (key "dummy_key"), dtu-12d
    _, mv = CheckpointHandler.load_mapper(os.path.join(_G, "f2_mapper-steps-7_view.pt"), "view",
                                          cam_mins=torch.from_numpy(exp["cam_mins"]),
                                          cam_maxs=torch.from_numpy(exp["cam_maxs"]))
    with torch.no_grad():
        wv, bv = mv(t, lay, torch.from_numpy(exp["view_scaled"]))
    assert torch.allclose(wv, torch.from_numpy(exp["word_view"]), atol=1e-6)
    assert torch.allclose(bv, torch.from_numpy(exp["bypass_view"]), atol=1e-5)
    assert torch.equal(mv.encoder.w, torch.from_numpy(exp["w_view"]))


def test_f2_load_learned_embed_in_clip():
    """checkpoint_handler.py:232-267 on a learned_embeds file in the reference's format"""
    from view_neti_amd.compat.checkpoint_handler import TextEncoderWeights
    exp = np.load(os.path.join(_G, "f2_expected.npz"))
    tok = HashTokenizer(96)
    E0 = torch.randn(96, 32)
    enc = TextEncoderWeights({TextEncoderWeights.KEY: E0.clone()})
    tokens, ids = CheckpointHandler.load_learned_embed_in_clip(os.path.join(_G, "f2_learned_embeds-steps-7.bin"), enc, tok)
    assert tokens == list(exp["emb_tokens"]) and tokens[-1] == "<obj>" and ids == [96, 97, 98] and len(tok) == 99
    E = enc.get_input_embeddings().weight
    assert E.shape == (99, 32) and torch.equal(E[:96], E0)
    assert torch.equal(E[96:], torch.from_numpy(exp["emb_values"]))
    with pytest.raises(ValueError, match="already contains"):
        CheckpointHandler.load_learned_embed_in_clip(os.path.join(_G, "f2_learned_embeds-steps-7.bin"), enc, tok)


def test_f2_diffusers_layout_safetensors(tmp_path):
    """compat/sd_weights.py: a diffusers-layout checkpoint directory (unet/, vae/, text_encoder/ *.safetensors) is read
    by state-dict key; the post-0.14 VAE attention names map back; a missing tensor is an error."""
    from safetensors.torch import save_file
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.compat import sd_weights
    cfg = sc.tiny()
    uw, cw = synth.unet_weights(cfg.unet, device="cpu"), synth.clip_weights(cfg.clip, device="cpu")
    vw = dict(synth.vae_weights(cfg.vae, device="cpu"))
    vw.update(synth.vae_decoder_weights(cfg.vae, device="cpu"))
    new_names = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    vae_file = {}
    for k, v in vw.items():
        for old, new in new_names.items():
            k = k.replace(f"attentions.0.{old}.", f"attentions.0.{new}.")
        vae_file[k] = v.contiguous()
    assert any(".to_q." in k for k in vae_file)
    for sub, d in (("unet", uw), ("vae", vae_file), ("text_encoder", cw)):
        os.makedirs(tmp_path / sub)
        items = sorted(d.items())
        half = len(items) // 2   # two shards, like large checkpoints
        save_file({k: v.contiguous() for k, v in items[:half]}, str(tmp_path / sub / "model-00001.safetensors"))
        save_file({k: v.half().contiguous() for k, v in items[half:]}, str(tmp_path / sub / "model-00002.safetensors"))
    u2, v2, c2, synthetic = sd_weights.load_sd_weights(cfg, str(tmp_path), device="cpu")
    assert synthetic is False
    for got, want, shapes in ((u2, uw, sc.unet_shapes(cfg.unet)), (v2, vw, sc.vae_encoder_shapes(cfg.vae)),
                              (c2, cw, sc.clip_text_shapes(cfg.clip))):
        assert set(got) == set(shapes)
        for k in shapes:
            assert got[k].dtype == torch.float32 and torch.allclose(got[k], want[k].float(), atol=2e-3, rtol=1e-3), k
    dec, synthetic = sd_weights.load_vae_decoder_weights(cfg, str(tmp_path), device="cpu")
    assert synthetic is False and set(dec) == set(sc.vae_decoder_shapes(cfg.vae))
    os.remove(tmp_path / "unet" / "model-00001.safetensors")
    with pytest.raises(KeyError, match="missing from checkpoint"):
        sd_weights.load_sd_weights(cfg, str(tmp_path), device="cpu")


def test_legacy_mapper_module_matches_reference_and_round_trips(tmp_path):
    """SURVEY a5': compat NeTIMapper(arch_view_net=0) — the reference's dataclass default — against G9 (outputs of the real
    module), and through a checkpoint whose pickled NeTIPositionalEncoding carries the un-seeded frequencies."""
    from view_neti_amd.compat.neti_modules import NeTIPositionalEncoding
    f = np.load(os.path.join(_G, "g9_legacy_mapper.npz"))
    D = f["word"].shape[1]
    m = NeTIMapper("object", D, 128, 0.4, arch_view_net=0, placeholder_object_token="<obj>")
    assert m.legacy and isinstance(m.encoder, NeTIPositionalEncoding) and m.enc_dim == 160 and m.pe_dim == 2048
    assert type(m.encoder).__module__ == "models.positional_encoding"
    assert torch.allclose(m.input_layer.weight.norm(dim=1), torch.ones(160), atol=1e-5)  # anchor rows are unit vectors
    n768 = sum(v.numel() for v in NeTIMapper("object", 768, 128, 0.4, arch_view_net=0).mapper_state().values())
    assert n768 == int(f["n_params_768"]) == 563616
    m.encoder.w = torch.from_numpy(f["w_pe"])
    sd = {k[3:]: torch.from_numpy(v) for k, v in f.items() if k.startswith("sd.")}
    sd["input_layer.weight"] = m.encoder.init_layer(10, 16)
    assert set(sd) == set(m.mapper_state())
    m.load_state_dict(sd, strict=False)
    t, lay = torch.from_numpy(f["t"]), torch.from_numpy(f["l"])
    with torch.no_grad():
        w, b = m(t, lay)
    assert torch.allclose(w, torch.from_numpy(f["word"]), atol=1e-5) and torch.allclose(b, torch.from_numpy(f["bypass"]), atol=1e-4)
    with pytest.raises(NotImplementedError):
        NeTIMapper("view", D, 64, 0.4, arch_view_net=0)
    # checkpoint round trip: the frequencies survive only through the pickled encoder
    cfg = C.RunConfig(data=C.DataConfig(train_data_dir="x", placeholder_object_token="<obj>"),
                      model=C.ModelConfig(word_embedding_dim=D, target_norm_object=0.4, use_nested_dropout=False))
    assert cfg.model.arch_view_net == 0 and cfg.model.arch_mlp_hidden_dims == 128  # the reference's defaults
    h = CheckpointHandler(cfg, [], [], ["<obj>"], [96], tmp_path)
    h.save_mapper({96: m}, None, "mapper-steps-1.pt")
    _, lookup = CheckpointHandler.load_mapper(tmp_path / "mapper-steps-1_object.pt", "object", ["<obj>"], [96])
    m2 = lookup[96]
    assert m2.legacy and torch.equal(m2.encoder.w, m.encoder.w)
    with torch.no_grad():
        w2, b2 = m2(t, lay)
    assert torch.equal(w2, w) and torch.equal(b2, b)


def test_bench_gpus_n_self_launches_to_the_ranks():
    """`python bench.py --gpus 2` (the driver's invocation, no launcher, no WORLD_SIZE) must start 2 ranks through
    torch.distributed.run on 127.0.0.1; in the GPU-less container each rank then fails at device selection — i.e. it got
    past the rendezvous-less launch instead of exiting with "launch with torch.distributed.run"."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered by tests/test_bench_gpu.py::test_bench_self_launches_two_ranks on a GPU box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--model", "tiny", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in r.stderr
    assert r.stderr.count("No HIP GPUs are available") >= 2, r.stderr[-1500:]
    assert "local_rank" in r.stderr  # torch.distributed.run's failure report: the ranks existed
