"""Generate the golden fixtures under tests/golden/ by running the REAL reference modules.

Runs only in the build container (needs /root/reference); the GPU box sees just the committed
.npz files.  The reference is pure Python but hard-codes `.cuda()` and imports packages that are
absent here (ipdb, torchvision, diffusers); those imports are satisfied by inert stub modules and
`.cuda()` is patched to the identity — no reference source is copied or modified.

What is importable and therefore pinned by the real code:
  G1  models.positional_encoding.FourierPositionalEncodingNDims       (w, forward)
  G2  models.neti_mapper.NeTIMapper, embedding_type='object', arch 15  (state_dict, fwd, param grads)
  G3  models.neti_mapper.NeTIMapper, embedding_type='view', dtu-12d    (synthetic calibration files)
  G4  models.net_clip_text_embedding.NeTICLIPTextEmbeddings            (overwrite + position add)
  G5  models.xti_attention_processor.XTIAttenProc                      (None / tensor / dict contexts)
  G6  transformers' CLIPTextModel (third-party stack the reference subclasses) with random weights
  G7  models.neti_clip_text_encoder.NeTICLIPTextModel(batch=NeTIBatch), bypass included — imported over a small
      transformers-4.27 shim built from the installed library's own layers (install_transformers_4_27_shim)
  G8  NeTIMapper.scale_m1_1, utils.utils.num_to_string/string_to_num
  G9  the legacy (arch_view_net <= 14) object mapper;  F2  checkpoints pickled by the reference's own classes
  G10 sd_pipeline_call.sd_pipeline_call — the REAL denoising loop (:73-98) driven with a duck-typed pipeline
      (oracle/toys.py: toy UNet, a scheduler object over oracle/sd_ref.sampler_step): call order, per-step
      `prompt_embeds[i]`, the CFG formula, the scheduler-step sequence, output_type handling
  G11 prompt_manager.PromptManager.embed_prompt (:43-101) over the G7 text encoder: the T x 16 context dicts
Not pinnable: UNet2DConditionModel / AutoencoderKL / the diffusers schedulers (diffusers is absent from the reference
tree and from this image) — oracle/sd_ref.py restates their published 0.14 architecture, PARITY UNPINNED.

Each fixture is immediately cross-checked against oracle/sd_ref.py; the script fails if the
restatement and the reference disagree.

Usage:  python oracle/make_golden.py
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Identity:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


def install_reference():
    # transformers' CLIP must be imported BEFORE a fake torchvision exists
    import transformers.models.clip.modeling_clip  # noqa: F401
    _stub("ipdb", set_trace=lambda *a, **k: None)
    tv = _stub("torchvision")
    names = ["Compose", "RandomApply", "ColorJitter", "RandomGrayscale", "GaussianBlur", "RandomRotation",
             "RandomResizedCrop", "RandomHorizontalFlip", "Resize", "ToTensor", "Normalize", "CenterCrop",
             "InterpolationMode"]
    tvt = _stub("torchvision.transforms", **{n: _Identity for n in names})
    tv.transforms = tvt
    d = _stub("diffusers")
    dm = _stub("diffusers.models")
    dc = _stub("diffusers.models.cross_attention", CrossAttention=object)
    d.models = dm
    dm.cross_attention = dc
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def install_transformers_4_27_shim():
    """models/neti_clip_text_encoder.py:9,103,111 subclasses transformers-4.27 internals that transformers 5.x no
    longer has.  Give the import what it asks for, built ON TOP OF the installed library's own layers (no reference
    code involved): `_expand_mask`, a `CLIPTextTransformer` base whose only inherited member the reference uses is
    `_build_causal_attention_mask` (4.27: finfo.min above the diagonal, shape (B,1,L,L)), and a `CLIPEncoder`
    whose forward accepts the 4.27 keyword set and adds the causal mask to the padding mask before running the
    installed CLIPEncoderLayer stack."""
    import transformers.models.clip.modeling_clip as mc
    from transformers.modeling_outputs import BaseModelOutput

    if hasattr(mc, "CLIPTextTransformer"):
        return

    def _expand_mask(mask, dtype, tgt_len=None):
        bsz, src_len = mask.size()
        tgt_len = tgt_len if tgt_len is not None else src_len
        inv = 1.0 - mask[:, None, None, :].expand(bsz, 1, tgt_len, src_len).to(dtype)
        return inv.masked_fill(inv.to(torch.bool), torch.finfo(dtype).min)

    class CLIPTextTransformer(torch.nn.Module):
        def __init__(self, config):
            super().__init__()

        def _build_causal_attention_mask(self, bsz, seq_len, dtype):
            mask = torch.empty(bsz, seq_len, seq_len, dtype=dtype)
            mask.fill_(torch.tensor(torch.finfo(dtype).min))
            mask.triu_(1)
            return mask.unsqueeze(1)

    base_encoder = mc.CLIPEncoder

    class CLIPEncoder(base_encoder):
        def forward(self, inputs_embeds, attention_mask=None, causal_attention_mask=None, output_attentions=None,
                    output_hidden_states=None, return_dict=None, **kwargs):
            mask = causal_attention_mask
            if attention_mask is not None:
                mask = attention_mask if mask is None else mask + attention_mask
            h = inputs_embeds
            for layer in self.layers:
                h = layer(h, mask, **kwargs)
            return BaseModelOutput(last_hidden_state=h, hidden_states=None, attentions=None)

    mc._expand_mask = _expand_mask
    mc.CLIPTextTransformer = CLIPTextTransformer
    mc.CLIPEncoder = CLIPEncoder


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def close(a, b, tol, what):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-30
    assert err <= tol * max(1.0, ref), f"{what}: oracle vs reference max abs err {err} (ref max {ref})"
    print(f"  ok {what}: max abs err {err:.3e}")


import contextlib


@contextlib.contextmanager
def synthetic_calibration(seed=11):
    """a temp CWD holding 64 seeded 3x4 camera matrices where constants.py:13 expects the DTU calibration files"""
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            cal = os.path.join("data", "dtu", "Calibration", "cal18")
            os.makedirs(cal)
            rng = np.random.RandomState(seed)
            mats = rng.randn(64, 3, 4) * np.array([[1e3, 1e3, 1e3, 1e5]])
            for i in range(64):
                np.savetxt(os.path.join(cal, f"pos_{i + 1:03d}.txt"), mats[i])
            yield mats
        finally:
            os.chdir(cwd)


def g_text_encoder_bypass(R, NeTIMapper, NeTIBatch, PESigmas):
    """models/neti_clip_text_encoder.py:15-225 run for real (NeTICLIPTextModel(batch=NeTIBatch)) on a tiny CLIP
    config with an object mapper and a dtu-12d view mapper, constrained and unconstrained bypass; the oracle's
    neti_text_encoder must reproduce both returned hidden states."""
    install_transformers_4_27_shim()
    from transformers import CLIPTextConfig
    from models.neti_clip_text_encoder import NeTICLIPTextModel
    from training.dataset import TextualInversionDataset as TID
    from view_neti_amd import sd_config as sc
    Dh, V = 32, 96
    tc = CLIPTextConfig(vocab_size=V, hidden_size=Dh, max_position_embeddings=77, num_hidden_layers=2,
                        num_attention_heads=2, intermediate_size=64, hidden_act="quick_gelu")
    tc._attn_implementation = "eager"
    my_cfg = sc.CLIPTextConfig(vocab_size=V, hidden_size=Dh, num_layers=2, num_heads=2, intermediate_size=64,
                               act="quick_gelu")
    torch.manual_seed(21)
    model = NeTICLIPTextModel(tc).eval()
    tm = model.text_model
    gen = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for prm in tm.parameters():  # O(1) activations through the stack, informative LayerNorm affine
            prm.copy_(torch.randn(prm.shape, generator=gen) * (0.25 if prm.dim() > 1 else 0.1))
        for n, prm in tm.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                prm.add_(1.0)
    cw = {"text_model." + k: v.detach().clone() for k, v in tm.state_dict().items() if "position_ids" not in k}
    obj_id, B = 90, 4
    ids = torch.randint(0, 88, (B, 77), generator=torch.Generator().manual_seed(3))
    pos_obj, pos_view = [5, 9, 2, 30], [7, 3, 11, 31]
    arrays = {"cw." + k: v for k, v in cw.items()}
    with synthetic_calibration() as mats:
        toks, _ = TID.dtu_generate_dset_cam_tokens_params()
        cams = [0, 8, 13, 22]
        view_tokens = [toks[c] for c in cams]
        view_ids = [91 + i for i in range(len(cams))]
        for b in range(B):
            ids[b, pos_obj[b]] = obj_id
            ids[b, pos_view[b]] = view_ids[(b * 3) % 4]
        ph_view = torch.tensor([view_ids[(b * 3) % 4] for b in range(B)])
        t = torch.tensor([3, 700, 42, 999])
        lay = torch.tensor([4, 4, 4, 4])
        for tag, unc, with_view in (("obj", False, False), ("objview", False, True), ("objview_unc", True, True)):
            torch.manual_seed(31)
            mo = NeTIMapper(embedding_type="object", output_dim=Dh, arch_mlp_hidden_dims=64, use_nested_dropout=False,
                            norm_scale=torch.tensor(0.4), pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0),
                            output_bypass=True, arch_view_net=15, arch_view_disable_tl=False,
                            bypass_unconstrained=unc, output_bypass_alpha=0.3).eval()
            torch.manual_seed(32)
            mv = NeTIMapper(embedding_type="view", output_dim=Dh, use_nested_dropout=False,
                            norm_scale=torch.tensor(0.35),
                            pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0, sigma_dtu12=0.5), output_bypass=True,
                            placeholder_view_tokens=list(view_tokens), placeholder_view_token_ids=list(view_ids),
                            arch_view_net=15, arch_view_disable_tl=False, bypass_unconstrained=unc,
                            output_bypass_alpha=0.15).eval()
            g2 = torch.Generator().manual_seed(33)
            with torch.no_grad():
                for m in (mo, mv):
                    for n, prm in m.named_parameters():
                        if n != "encoder.w":
                            prm.add_(0.1 * torch.randn(prm.shape, generator=g2))
            tm.embeddings.set_mapper({obj_id: mo}, mv if with_view else None, device="cpu")
            batch = NeTIBatch(input_ids=ids.clone(), input_ids_placeholder_object=torch.full((B,), obj_id),
                              input_ids_placeholder_view=ph_view.clone() if with_view else torch.full((B,), -1),
                              timesteps=t, unet_layers=lay)
            with torch.no_grad():
                out, out_b = model(batch=batch)
            last, last_b = out.last_hidden_state, out_b.last_hidden_state
            sdo = {k: v.detach().clone() for k, v in mo.state_dict().items() if k != "encoder.w"}
            sdv = {k: v.detach().clone() for k, v in mv.state_dict().items() if k != "encoder.w"}
            params = torch.stack([mv.view_tokenid_2_view_params[i.item()] for i in ph_view]).float()
            scaled = NeTIMapper.scale_m1_1(params, mv.cam_mins, mv.cam_maxs)
            with torch.no_grad():
                wo, bo = R.mapper_forward(sdo, R.fourier_w([0.03, 2.0]), t, lay, 0.4)
                wv = bv = None
                if with_view:
                    wv, bv = R.mapper_forward(sdv, R.fourier_w([0.03, 2.0] + [0.5] * 12), t, lay, 0.35,
                                              view_params=scaled)
                m_last, m_b = R.neti_text_encoder(cw, my_cfg, ids, torch.full((B,), obj_id), wo, bo, unc, 0.3,
                                                  ph_view if with_view else None, wv, bv, unc, 0.15)
            close(m_last, last, 2e-5, f"G7 text encoder [{tag}] last_hidden_state")
            close(m_b, last_b, 2e-5, f"G7 text encoder [{tag}] last_hidden_state_with_bypass")
            assert (last - last_b).abs().max() > 1e-2
            arrays.update({f"{tag}.last": last, f"{tag}.last_bypass": last_b, f"{tag}.pooled": out.pooler_output,
                           f"{tag}.pooled_bypass": out_b.pooler_output})
            if tag != "objview":  # same mapper weights in the two objview cases except the flag
                arrays.update({f"{tag}.sdo.{k}": v for k, v in sdo.items()})
            if tag == "objview":
                arrays.update({f"objview.sdo.{k}": v for k, v in sdo.items()})
                arrays.update({f"sdv.{k}": v for k, v in sdv.items()})
                arrays.update(view_scaled=scaled)
        # the plain input_ids= path of the same module (sd_pipeline_call.py:35-39 uses it for the negative prompt)
        tm.embeddings.set_mapper({obj_id: mo}, None, device="cpu")
        with torch.no_grad():
            plain, none = model(input_ids=ids)
        assert none is None
        close(R.clip_plain(cw, my_cfg, ids), plain.last_hidden_state, 2e-5, "G7 text encoder plain input_ids path")
        arrays.update(plain_last=plain.last_hidden_state)
    save("g7_text_encoder_bypass", ids=ids, ph_view=ph_view, obj_id=np.array(obj_id), timesteps=t, layers=lay,
         alpha_obj=np.array(0.3), alpha_view=np.array(0.15), norm_obj=np.array(0.4), norm_view=np.array(0.35), **arrays)


def g6_config_dumps():
    """training/config.py:11-293 run for real: RunConfig() and the three shipped train YAMLs decoded the way
    pyrallis does (field-type coercion, nested dataclasses), then the post-`__post_init__` state dumped to plain
    dicts.  The fixture keeps the raw YAML mapping (input) next to the dump (expected) so the CPU test needs no
    reference file.  pyrallis is absent: the few lines below rebuild its decode over the REFERENCE dataclasses."""
    import dataclasses
    import json
    import typing
    from pathlib import Path
    import yaml
    import training.config as rc

    def build(cls, d):
        hints = typing.get_type_hints(cls)
        kw = {}
        for f in dataclasses.fields(cls):
            if f.name not in d:
                continue
            v, tp = d[f.name], hints[f.name]
            if dataclasses.is_dataclass(tp):
                v = build(tp, v)
            elif tp is Path or (typing.get_origin(tp) is typing.Union and Path in typing.get_args(tp) and
                                str not in typing.get_args(tp)):
                v = None if v is None else Path(v)
            elif tp in (int, float) or (typing.get_origin(tp) is typing.Union and
                                        set(typing.get_args(tp)) <= {int, float, type(None)}):
                base = tp if tp in (int, float) else [a for a in typing.get_args(tp) if a is not type(None)][0]
                v = None if v is None else base(v)
            elif tp is str:  # pyrallis decodes by annotation: `dtu_lighting: 3` in a YAML becomes '3'
                v = None if v is None else str(v)
            kw[f.name] = v
        return cls(**kw)

    def plain(o):
        if dataclasses.is_dataclass(o) and not isinstance(o, type):
            return {f.name: plain(getattr(o, f.name)) for f in dataclasses.fields(o)}
        if isinstance(o, Path):
            return str(o)
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        if isinstance(o, type):  # utils/types.py:21-24 default typo `= float`
            return None
        return o

    cases = {}
    # the dataclass default of data.train_data_dir is a required field: give the one mandatory value
    sources = {"default": {"data": {"train_data_dir": "data/x"}}}
    for name in ("train", "train_m3", "train_m3_88scenes"):
        with open(os.path.join(REF, "input_configs", name + ".yaml")) as f:
            sources[name] = yaml.safe_load(f)
    # exp-key switches (config.py:151-178) on top of train.yaml
    for tag, over in (("train_keys_a", {"pe_sigma_exp_key": 4, "pe_t_exp_key": 2, "pe_l_exp_key": 1}),
                      ("train_keys_b", {"pe_sigma_exp_key": 0, "pe_t_exp_key": 3})):
        src = json.loads(json.dumps(sources["train"]))
        src["model"].update(over)
        sources[tag] = src
    for name, src in sources.items():
        src = json.loads(json.dumps(src))
        rec = {"input": src}
        try:
            cfg = build(rc.RunConfig, src)
            rec["dump"] = plain(cfg)
        except (AssertionError, ValueError, TypeError) as e:
            rec["raises"] = type(e).__name__
            # the mode-3 YAMLs need the CLI-supplied super-category list (config.py:274): add it and dump that too
            if src.get("learnable_mode") == 3 and not src["data"].get("super_category_object_tokens"):
                src2 = json.loads(json.dumps(src))
                if not src2["data"].get("placeholder_object_tokens"):  # train.yaml leaves these to the CLI as well
                    src2["data"]["placeholder_object_tokens"] = list(
                        src2.get("eval", {}).get("eval_placeholder_object_tokens") or ["<object>"])
                n = len(src2["data"]["placeholder_object_tokens"])
                src2["data"]["super_category_object_tokens"] = ["object"] * n
                try:
                    rec["input_completed"] = src2
                    rec["dump_completed"] = plain(build(rc.RunConfig, src2))
                except (AssertionError, ValueError, TypeError) as e2:
                    rec["raises_completed"] = type(e2).__name__
        cases[name] = rec
        print(f"  G6 {name}: " + ("dump" if "dump" in rec else f"raises {rec['raises']}" +
                                  (" (completed dump ok)" if "dump_completed" in rec else "")))
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "g6_config_dumps.json")
    with open(path, "w") as f:
        json.dump(cases, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def g7_dataset_statics():
    """training/dataset.py:321-408,455-522 statics run for real: train-view splits, file-name <-> camera index,
    lighting / index filters, token <-> parameter round trip, the calibration reader (on the synthetic files)."""
    import json
    from pathlib import Path
    from training.dataset import TextualInversionDataset as TID
    rec = {"train_idxs": {}}
    for k in (0, 1, 3, 6, 9, -1, -2, -3):
        rec["train_idxs"][str(k)] = list(TID.dtu_get_train_idxs(k))
    try:
        TID.dtu_get_train_idxs(2)
        rec["train_idxs_2_raises"] = None
    except NotImplementedError:
        rec["train_idxs_2_raises"] = "NotImplementedError"
    names = [TID.dtu_cam_and_lighting_to_fname(c, l) for c in (0, 7, 24, 48) for l in ("3", "max")]
    rec["fnames"] = names
    rec["cam_info"] = [list(TID.dtu_cam_info_from_fname(Path("x") / n)) for n in names]
    paths = [Path("scan1") / n for n in names]
    rec["filter_lighting_3"] = [str(p) for p in TID.dtu_filter_fnames_lighting(paths, "3")]
    rec["filter_idx"] = [str(p) for p in TID.dtu_filter_image_paths_from_idx(list(reversed(paths)), [24, 0, 48])]
    with synthetic_calibration() as mats:
        toks, params = TID.dtu_generate_dset_cam_tokens_params()
        rec["calib"] = mats.tolist()
        rec["tokens"] = {str(k): toks[k] for k in sorted(toks)}
        rec["params"] = {str(k): params[k].flatten().tolist() for k in sorted(params)}
        back = {}
        for k in (0, 13, 63):
            p, key = TID.dtu_token_to_cam_params(toks[k], cam_idx_as_int=True)
            back[str(k)] = {"params": p.tolist(), "key": key}
        rec["token_to_params"] = back
        novel = torch.tensor(mats[5]).float() * 1.5
        rec["novel_token"] = TID.dtu_cam_params_to_token(novel)
        rec["novel_params"] = novel.flatten().tolist()
    path = os.path.join(OUT, "g7_dataset_statics.json")
    with open(path, "w") as f:
        json.dump(rec, f)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def f2_reference_checkpoints(R, NeTIMapper, PESigmas):
    """SURVEY f2 / checkpoint_handler.py:34-97: mapper checkpoints and a learned-embeds file laid out exactly as the
    reference's CheckpointHandler.save_model writes them, holding the REFERENCE's own objects: `state_dict()` of its
    NeTIMapper and the pickled `models.positional_encoding.FourierPositionalEncodingNDims` instance.  (pyrallis is
    absent, so `pyrallis.encode(cfg)` is the plain-dict dump of the reference RunConfig used for G6; the file is
    assembled by hand in the documented layout.)  On a CUDA box `nn.Parameter(w).cuda()` leaves `encoder.w` a plain
    tensor attribute outside state_dict (App. C Q2); the no-op `.cuda()` of this harness would register it, so it is
    moved back to what a real run pickles."""
    import dataclasses
    from pathlib import Path
    import yaml
    import training.config as rc
    from training.dataset import TextualInversionDataset as TID
    D = 32

    def as_cuda_run(m):
        enc = m.encoder
        if "w" in enc._parameters:
            w = enc._parameters.pop("w").detach().clone()
            enc.__dict__["w"] = w
        return {k: v.detach().clone() for k, v in m.state_dict().items() if k != "encoder.w"}

    def plain(o):
        if dataclasses.is_dataclass(o) and not isinstance(o, type):
            return {f.name: plain(getattr(o, f.name)) for f in dataclasses.fields(o)}
        if isinstance(o, Path):
            return str(o)
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return None if isinstance(o, type) else o

    with open(os.path.join(REF, "input_configs", "train_m3.yaml")) as f:
        src = yaml.safe_load(f)
    cfg = rc.RunConfig(
        learnable_mode=2, log=rc.LogConfig(exp_name="f2", exp_dir=Path("results")),
        data=rc.DataConfig(train_data_dir=Path("data/dtu/Rectified/scan114"), placeholder_object_token="<obj>",
                           camera_representation="dtu-12d", dtu_subset=6, dataloader_num_workers=0),
        model=rc.ModelConfig(word_embedding_dim=D, arch_mlp_hidden_dims=64, use_nested_dropout=False,
                             normalize_view_mapper_output=True, arch_view_net=15, arch_view_disable_tl=False,
                             pe_sigma_exp_key=2, output_bypass_alpha_view=5, output_bypass_alpha_object=5,
                             pe_sigmas=dict(src["model"]["pe_sigmas"]), target_norm_object=0.4, target_norm_view=0.35))
    enc_cfg = plain(cfg)
    gen = torch.Generator().manual_seed(41)
    t = torch.tensor([10.0, 500.0, 999.0, 3.0])
    lay = torch.tensor([0.0, 7.0, 15.0, 2.0])
    with synthetic_calibration() as mats:
        toks, _ = TID.dtu_generate_dset_cam_tokens_params()
        cams = [25, 22, 28, 40, 44, 48]
        view_tokens, view_ids = [toks[c] for c in cams], [91 + i for i in range(len(cams))]
        torch.manual_seed(42)
        mo = NeTIMapper(embedding_type="object", output_dim=D, arch_mlp_hidden_dims=64, use_nested_dropout=False,
                        norm_scale=torch.tensor(0.4), pe_sigmas=cfg.model.pe_sigmas, output_bypass=True,
                        arch_view_net=15, arch_view_disable_tl=False, bypass_unconstrained=False,
                        output_bypass_alpha=5, placeholder_object_token="<obj>").eval()
        mv = NeTIMapper(embedding_type="view", output_dim=D, use_nested_dropout=False, norm_scale=torch.tensor(0.35),
                        pe_sigmas=cfg.model.pe_sigmas, output_bypass=True, placeholder_view_tokens=list(view_tokens),
                        placeholder_view_token_ids=list(view_ids), arch_view_net=15, arch_view_disable_tl=False,
                        bypass_unconstrained=False, output_bypass_alpha=5).eval()
        with torch.no_grad():
            for m in (mo, mv):
                for n, prm in m.named_parameters():
                    if n != "encoder.w":
                        prm.add_(0.1 * torch.randn(prm.shape, generator=gen))
            ids = torch.tensor([view_ids[0], view_ids[3], view_ids[5], view_ids[1]])
            out_o = mo(timestep=t, unet_layer=lay, input_ids_placeholder_view=None, truncation_idx=None)
            out_v = mv(timestep=t, unet_layer=lay, input_ids_placeholder_view=ids, truncation_idx=None)
            params = torch.stack([mv.view_tokenid_2_view_params[i.item()] for i in ids]).float()
            scaled = NeTIMapper.scale_m1_1(params, mv.cam_mins, mv.cam_maxs)
        sdo, sdv = as_cuda_run(mo), as_cuda_run(mv)
        os.makedirs(OUT, exist_ok=True)
        torch.save({"cfg": enc_cfg, "mappers": {90: {"state_dict": sdo, "encoder": mo.encoder,
                                                    "placeholder_object_token": "<obj>"}}},
                   os.path.join(OUT, "f2_mapper-steps-7_object.pt"))
        torch.save({"cfg": enc_cfg, "mappers": {"dummy_key": {"state_dict": sdv, "encoder": mv.encoder,
                                                              "placeholder_object_token": "dummy"}}},
                   os.path.join(OUT, "f2_mapper-steps-7_view.pt"))
        # learned_embeds-*.bin: {token: Tensor(D,) cpu fp32}, view tokens first, then object tokens (:40-55)
        emb = {tok: torch.randn(D, generator=gen) for tok in view_tokens[:2] + ["<obj>"]}
        torch.save(emb, os.path.join(OUT, "f2_learned_embeds-steps-7.bin"))
        save("f2_expected", t=t, l=lay, word_obj=out_o.word_embedding, bypass_obj=out_o.bypass_output,
             word_view=out_v.word_embedding, bypass_view=out_v.bypass_output, view_scaled=scaled, cam_mins=mv.cam_mins,
             cam_maxs=mv.cam_maxs, emb_tokens=np.array(list(emb)), emb_values=torch.stack(list(emb.values())),
             sigma_dtu12=np.array(cfg.model.pe_sigmas.sigma_dtu12), w_view=mv.encoder.w, w_obj=mo.encoder.w)
    for f in ("f2_mapper-steps-7_object.pt", "f2_mapper-steps-7_view.pt", "f2_learned_embeds-steps-7.bin"):
        print(f"wrote {os.path.join(OUT, f)} ({os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB)")


def g9_legacy_mapper(R, NeTIMapper, PESigmas):
    """SURVEY a5' / G9: the reference's DATACLASS-DEFAULT object mapper (arch_view_net = 0, arch_view_disable_tl = True,
    use_positional_encoding = 1, arch_mlp_hidden_dims = 128: training/config.py:89-130) — NeTIPositionalEncoding
    (2048-d normalised sin/cos of raw (t, l)) -> anchor-initialised input_layer(2048 -> 160) -> MLP(h) -> 2D.  Captures
    the encoder frequencies, the anchor initialisation (row norms 1), the outputs and the parameter gradients.  A small
    output width keeps the fixture small; the parameter count at D = 768 is asserted from shapes: 563 616 without
    `encoder.w` (SURVEY's 565 664 counts the 1024x2 frequencies, which a CUDA run keeps out of state_dict, App. C Q2)."""
    from models.positional_encoding import NeTIPositionalEncoding
    D, h = 24, 128
    torch.manual_seed(77)
    m = NeTIMapper(embedding_type="object", output_dim=D, arch_mlp_hidden_dims=h, use_nested_dropout=False,
                   norm_scale=torch.tensor(0.4), pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0), output_bypass=True,
                   bypass_unconstrained=False, output_bypass_alpha=0.2, placeholder_object_token="<obj>").eval()
    assert isinstance(m.encoder, NeTIPositionalEncoding) and m.arch_view_net == 0
    w_pe = m.encoder.w.detach().clone()
    init = m.input_layer.weight.detach().clone()
    close(R.neti_pe_init_layer(w_pe), init, 1e-6, "G9 input_layer anchor initialisation")
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k != "encoder.w"}
    n768 = sum(v.numel() for k, v in sd.items() if "output_layer" not in k) + (128 * 2 * 768 + 2 * 768)
    assert n768 == 563616 and n768 + w_pe.numel() == 565664, n768
    gen = torch.Generator().manual_seed(78)
    for k in sd:
        if k != "input_layer.weight":  # stays the anchor initialisation: the test rebuilds it from w_pe (1.3 MB saved)
            sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=gen)
    m.load_state_dict({**sd, **({"encoder.w": m.state_dict()["encoder.w"]} if "encoder.w" in m.state_dict() else {})})
    t = torch.tensor([10.0, 500.0, 999.0, 3.0])
    lay = torch.tensor([0.0, 7.0, 15.0, 2.0])
    out = m(timestep=t, unet_layer=lay, input_ids_placeholder_view=None, truncation_idx=None)
    gw = torch.randn(out.word_embedding.shape, generator=gen)
    gb = torch.randn(out.bypass_output.shape, generator=gen)
    ((out.word_embedding * gw).sum() + (out.bypass_output * gb).sum()).backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if k != "encoder.w" and p.grad is not None}
    po = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    wo, bo = R.mapper_forward_legacy(po, w_pe, t, lay, 0.4)
    ((wo * gw).sum() + (bo * gb).sum()).backward()
    close(R.neti_pe_encode(w_pe, t, lay), m.encoder.encode(t, lay), 1e-6, "G9 encode")
    close(wo, out.word_embedding, 1e-5, "G9 legacy word")
    close(bo, out.bypass_output, 1e-5, "G9 legacy bypass")
    for k in grads:
        close(po[k].grad, grads[k], 1e-4, f"G9 grad {k}")
    save("g9_legacy_mapper", w_pe=w_pe, init_row_norms=init.norm(dim=1), init_rows_0_17_159=init[[0, 17, 159]], t=t, l=lay,
         enc=m.encoder.encode(t, lay), word=out.word_embedding, bypass=out.bypass_output, gw=gw, gb=gb,
         n_params_768=np.array(n768), **{"sd." + k: v for k, v in sd.items() if k != "input_layer.weight"},
         **{"grad." + k: (v if v.numel() < 50000 else v[:, :16]) for k, v in grads.items()})


def _duck_pipeline(R, toys, cfg, kind, neg_embeds, log):
    """what `sd_pipeline_call` touches of a diffusers StableDiffusionPipeline, and nothing else"""
    from view_neti_amd import sd_config as sc  # noqa: F401

    class Scheduler:
        order = 1  # DPMSolverMultistepScheduler.order / DDIMScheduler.order

        def set_timesteps(self, n, device=None):
            self.ts = R.inference_timesteps(kind, n, cfg.ddpm.num_train_timesteps)
            self.timesteps = torch.tensor(self.ts)
            self.ac = R.alphas_cumprod(cfg.ddpm)
            self.i, self.m_prev = 0, None
            log.append(("set_timesteps", n))

        def scale_model_input(self, x, t):
            log.append(("scale_model_input", int(t)))
            return x

        def step(self, noise_pred, t, latents, **kw):
            assert int(t) == self.ts[self.i] and not kw
            m_prev = torch.zeros_like(latents) if self.m_prev is None else self.m_prev
            x, self.m_prev = R.sampler_step(cfg, kind, self.ac, self.ts, self.i, latents, noise_pred, m_prev)
            log.append(("step", int(t)))
            self.i += 1
            return type("SchedulerOutput", (), {"prev_sample": x})()

    class Tokenizer:
        model_max_length = 77

        def __call__(self, texts, padding=None, max_length=None, truncation=None, return_tensors=None):
            assert padding == "max_length" and max_length == 77 and truncation and return_tensors == "pt"
            log.append(("tokenizer", tuple(texts)))
            return type("Enc", (), {"input_ids": torch.full((len(texts), 77), 7)})()

    class TextEncoder:
        dtype = torch.float32

        def __call__(self, input_ids=None, attention_mask=None):
            assert attention_mask is None and input_ids.shape == (1, 77)
            log.append(("text_encoder", int(input_ids[0, 0])))
            return (neg_embeds,), None  # NeTICLIPTextModel returns (output, output_with_bypass); output[0] = last hidden

    class Bar:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self):
            log.append(("progress", 0))

    class Pipeline:
        vae_scale_factor = 8
        _execution_device = torch.device("cpu")

        def __init__(self):
            self.unet = toys.ToyUNet()
            self.scheduler = Scheduler()
            self.tokenizer = Tokenizer()
            self.text_encoder = TextEncoder()

        def prepare_latents(self, n, c, h, w, dtype, device, generator, latents):
            log.append(("prepare_latents", (n, c, h, w)))
            return latents

        def prepare_extra_step_kwargs(self, generator, eta):
            return {}

        def progress_bar(self, total=None):
            return Bar()

        def decode_latents(self, latents):
            log.append(("decode_latents", 0))
            return toys.toy_decode(latents).numpy()

        def numpy_to_pil(self, image):
            log.append(("numpy_to_pil", 0))
            return ["pil"] * len(image)

    return Pipeline()


def g10_sd_pipeline_call(R):
    """/root/reference/sd_pipeline_call.py:8-133 run for real on a duck-typed pipeline; the restatement
    R.sd_pipeline_call must issue the same UNet calls in the same order and end on the same latents."""
    from oracle import toys
    from view_neti_amd import sd_config as sc
    out = type("StableDiffusionPipelineOutput", (), {})

    def _init(self, images=None, nsfw_content_detected=None):
        self.images, self.nsfw_content_detected = images, nsfw_content_detected
    out.__init__ = _init
    dp = _stub("diffusers.pipelines")
    dps = _stub("diffusers.pipelines.stable_diffusion", StableDiffusionPipelineOutput=out, StableDiffusionPipeline=object)
    sys.modules["diffusers"].pipelines = dp
    dp.stable_diffusion = dps
    import sd_pipeline_call as ref_call  # the reference's own module
    assert ref_call.__file__.startswith(REF)
    cfg = sc.tiny()
    g = torch.Generator().manual_seed(101)
    D, L = 8, 5
    neg = torch.randn(1, L, D, generator=g)
    lat0 = torch.randn(1, 4, 8, 8, generator=g)
    arrays = dict(neg=neg, latents0=lat0, guidance=np.array(7.5))
    for tag, kind, steps, as_list in (("ddim_list", "ddim", 3, True), ("dpm_list", "dpm++2m", 4, True),
                                      ("dpm_single", "dpm++2m", 3, False)):
        def make_embeds():
            gg = torch.Generator().manual_seed(202)
            es = []
            for i in range(steps if as_list else 1):
                d = {"this_idx": 0, "_tag": i}
                for l in range(16):
                    d[f"CONTEXT_TENSOR_{l}"] = torch.randn(1, L, D, generator=gg)
                    d[f"CONTEXT_TENSOR_BYPASS_{l}"] = torch.randn(1, L, D, generator=gg)
                es.append(d)
            return es if as_list else es[0]
        log = []
        pipe = _duck_pipeline(R, toys, cfg, kind, neg, log)
        res = ref_call.sd_pipeline_call(pipe, make_embeds(), num_inference_steps=steps, guidance_scale=7.5,
                                        latents=lat0.clone(), output_type="latent")
        final = res.images
        assert res.nsfw_content_detected is None
        calls = list(pipe.unet.calls)
        # same again through the decode branch (output_type anything but "latent"/"pil")
        log2 = []
        pipe2 = _duck_pipeline(R, toys, cfg, kind, neg, log2)
        img = ref_call.sd_pipeline_call(pipe2, make_embeds(), num_inference_steps=steps, guidance_scale=7.5,
                                        latents=lat0.clone(), output_type="np", return_dict=False)[0]
        assert [e[0] for e in log2].count("decode_latents") == 1 and "numpy_to_pil" not in [e[0] for e in log2]
        # the restatement with the same toy pair
        mine_unet = toys.ToyUNet()
        img_m, x_m = R.sd_pipeline_call(cfg, None, None, make_embeds(), neg, lat0, kind, steps, 7.5,
                                        unet_fn=lambda x, t, hs: mine_unet(x, t, encoder_hidden_states=hs).sample,
                                        decode_fn=toys.toy_decode)
        close(x_m, final, 1e-6, f"G10 [{tag}] final latents")
        close(img_m, img, 1e-6, f"G10 [{tag}] decoded image")
        assert mine_unet.calls == calls, (mine_unet.calls, calls)
        # what the loop did, in order: per step scale_model_input -> (uncond, cond) UNet calls -> scheduler.step
        seq = [e for e in log if e[0] in ("scale_model_input", "step")]
        assert [e[0] for e in seq] == ["scale_model_input", "step"] * steps
        assert log[0] == ("tokenizer", ("",)) and log[1][0] == "text_encoder"
        arrays.update({f"{tag}.final": final, f"{tag}.image": img, f"{tag}.calls": np.array(calls, dtype=np.int64),
                       f"{tag}.steps": np.array(steps), f"{tag}.as_list": np.array(int(as_list)),
                       f"{tag}.sched_t": np.array([e[1] for e in seq if e[0] == "step"], dtype=np.int64)})
    arrays["kinds"] = np.array(["ddim", "dpm++2m", "dpm++2m"])
    arrays["tags"] = np.array(["ddim_list", "dpm_list", "dpm_single"])
    save("g10_sd_pipeline_call", **arrays)


def g11_prompt_manager(R, NeTIMapper, NeTIBatch, PESigmas):
    """/root/reference/prompt_manager.py:43-101 run for real: PromptManager.embed_prompt over the real NeTICLIPTextModel
    (the G7 construction) with an object mapper and a dtu-12d view mapper -> T dicts of 16 (context, bypass) pairs.  The
    restatement of the same conditioning (R.text_conditioning, one timestep at a time) must reproduce every tensor."""
    install_transformers_4_27_shim()
    from transformers import CLIPTextConfig
    from models.neti_clip_text_encoder import NeTICLIPTextModel
    from training.dataset import TextualInversionDataset as TID
    from view_neti_amd import sd_config as sc
    Dh, V = 32, 96
    tc = CLIPTextConfig(vocab_size=V, hidden_size=Dh, max_position_embeddings=77, num_hidden_layers=2,
                        num_attention_heads=2, intermediate_size=64, hidden_act="quick_gelu")
    tc._attn_implementation = "eager"
    my_cfg = sc.CLIPTextConfig(vocab_size=V, hidden_size=Dh, num_layers=2, num_heads=2, intermediate_size=64,
                               act="quick_gelu")
    torch.manual_seed(41)
    model = NeTICLIPTextModel(tc).eval()
    tm = model.text_model
    gen = torch.Generator().manual_seed(42)
    with torch.no_grad():
        for prm in tm.parameters():
            prm.copy_(torch.randn(prm.shape, generator=gen) * (0.25 if prm.dim() > 1 else 0.1))
        for n, prm in tm.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                prm.add_(1.0)
    cw = {"text_model." + k: v.detach().clone() for k, v in tm.state_dict().items() if "position_ids" not in k}
    obj_id = 90
    with synthetic_calibration():
        import prompt_manager as ref_pm  # the reference's own module (imports its constants.py)
        assert ref_pm.__file__.startswith(REF)
        toks, _ = TID.dtu_generate_dset_cam_tokens_params()
        cams = [0, 8, 13, 22]
        view_tokens = [toks[c] for c in cams]
        view_ids = [91 + i for i in range(len(cams))]
        torch.manual_seed(51)
        mo = NeTIMapper(embedding_type="object", output_dim=Dh, arch_mlp_hidden_dims=64, use_nested_dropout=False,
                        norm_scale=torch.tensor(0.4), pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0),
                        output_bypass=True, arch_view_net=15, arch_view_disable_tl=False,
                        bypass_unconstrained=False, output_bypass_alpha=0.3).eval()
        torch.manual_seed(52)
        mv = NeTIMapper(embedding_type="view", output_dim=Dh, use_nested_dropout=False, norm_scale=torch.tensor(0.35),
                        pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0, sigma_dtu12=0.5), output_bypass=True,
                        placeholder_view_tokens=list(view_tokens), placeholder_view_token_ids=list(view_ids),
                        arch_view_net=15, arch_view_disable_tl=False, bypass_unconstrained=False,
                        output_bypass_alpha=0.15).eval()
        g2 = torch.Generator().manual_seed(53)
        with torch.no_grad():
            for m in (mo, mv):
                for n, prm in m.named_parameters():
                    if n != "encoder.w":
                        prm.add_(0.1 * torch.randn(prm.shape, generator=g2))
        tm.embeddings.set_mapper({obj_id: mo}, mv, device="cpu")
        ids = torch.randint(0, 88, (1, 77), generator=torch.Generator().manual_seed(5))
        ids[0, 4], ids[0, 9] = view_ids[2], obj_id

        class Tok:
            model_max_length = 77

            def __call__(self, text, padding=None, max_length=None, return_tensors=None):
                assert padding == "max_length" and max_length == 77 and return_tensors == "pt"
                return type("Enc", (), {"input_ids": ids.clone()})()

        timesteps = [torch.tensor(999), torch.tensor(500), torch.tensor(20)]
        pm = ref_pm.PromptManager(tokenizer=Tok(), text_encoder=model, timesteps=timesteps,
                                  placeholder_view_token_ids=list(view_ids), placeholder_object_token_ids=[obj_id],
                                  torch_dtype=torch.float32)
        with torch.no_grad():
            embeds = pm.embed_prompt("<view_13> a photo of a <obj>", num_images_per_prompt=2)
        assert len(embeds) == 3 and all(e["this_idx"] == 0 for e in embeds)
        assert len(embeds[0]) == 1 + 32 and embeds[0]["CONTEXT_TENSOR_0"].shape == (2, 77, Dh)
        sdo = {k: v.detach().clone() for k, v in mo.state_dict().items() if k != "encoder.w"}
        sdv = {k: v.detach().clone() for k, v in mv.state_dict().items() if k != "encoder.w"}
        params = mv.view_tokenid_2_view_params[view_ids[2]].float()[None]
        scaled = NeTIMapper.scale_m1_1(params, mv.cam_mins, mv.cam_maxs)
        rows = list(range(16)) + [76]
        arrays = {"cw." + k: v for k, v in cw.items()}
        for ti, t in enumerate(timesteps):
            with torch.no_grad():
                hs = R.text_conditioning(cw, my_cfg, sdo, R.fourier_w([0.03, 2.0]), 0.4, ids, torch.tensor([obj_id]),
                                         t[None], alpha=0.3, view=dict(p=sdv, w_enc=R.fourier_w([0.03, 2.0] + [0.5] * 12),
                                                                       norm_scale=0.35, placeholder=torch.tensor([view_ids[2]]),
                                                                       params=scaled, alpha=0.15))
            for l in range(16):
                for key in (f"CONTEXT_TENSOR_{l}", f"CONTEXT_TENSOR_BYPASS_{l}"):
                    ref = embeds[ti][key]
                    assert torch.equal(ref[0], ref[1])  # .repeat(num_images_per_prompt, 1, 1)
                    close(hs[key], ref[:1], 2e-5, f"G11 t={int(t)} {key}") if l in (0, 15) else \
                        (lambda a, b: None)(0, 0)
                    assert (hs[key] - ref[:1]).abs().max() <= 2e-5 * max(1.0, ref.abs().max().item())
                    short = "b" if "BYPASS" in key else "c"
                    arrays[f"t{ti}.{short}{l}"] = ref[0, rows]
                    arrays[f"t{ti}.{short}{l}.sum"] = np.array([ref[0].double().sum().item(),
                                                               ref[0].double().abs().sum().item()])
    save("g11_prompt_manager", ids=ids, obj_id=np.array(obj_id), view_id=np.array(view_ids[2]), view_scaled=scaled,
         timesteps=np.array([int(t) for t in timesteps]), rows=np.array(rows), num_images=np.array(2),
         alpha_obj=np.array(0.3), alpha_view=np.array(0.15), norm_obj=np.array(0.4), norm_view=np.array(0.35),
         **{"sdo." + k: v for k, v in sdo.items()}, **{"sdv." + k: v for k, v in sdv.items()}, **arrays)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures can only be generated in the build container")
    install_reference()
    from oracle import sd_ref as R

    from models.positional_encoding import FourierPositionalEncodingNDims
    from models.neti_mapper import NeTIMapper
    from models.net_clip_text_embedding import NeTICLIPTextEmbeddings
    from models.xti_attention_processor import XTIAttenProc
    from utils.types import NeTIBatch, PESigmas
    from utils.utils import num_to_string, string_to_num

    # ---------------- G1: Fourier positional encoding --------------------------------------
    g = torch.Generator().manual_seed(7)
    for tag, sigmas in (("obj", [0.03, 2.0]), ("view", [0.03, 2.0] + [0.5] * 12)):
        enc = FourierPositionalEncodingNDims(dim=64, sigmas=sigmas, seed=0)
        x = torch.rand((8, len(sigmas)), generator=g) * 2 - 1
        y = enc(x)
        w = enc.w.detach()
        close(R.fourier_w(sigmas, 64, 0), w, 0, f"G1 w[{tag}]")
        close(R.fourier_encode(w, x), y, 1e-6, f"G1 fwd[{tag}]")
        save(f"g1_fourier_{tag}", sigmas=np.array(sigmas, dtype=np.float64), w=w, x=x, y=y)

    # ---------------- G2: object mapper, arch 15 --------------------------------------------
    for D in (768, 1024):
        torch.manual_seed(123)
        m = NeTIMapper(embedding_type="object", output_dim=D, arch_mlp_hidden_dims=64, use_nested_dropout=False,
                       norm_scale=torch.tensor(0.4), pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0),
                       output_bypass=True, arch_view_net=15, arch_view_disable_tl=False,
                       bypass_unconstrained=False, output_bypass_alpha=0.2, placeholder_object_token="<obj>")
        m.eval()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items() if k != "encoder.w"}
        # break the "all mappers start identical / tiny" symmetry so grads are informative
        gen = torch.Generator().manual_seed(5)
        for k in sd:
            sd[k] = sd[k] + 0.05 * torch.randn(sd[k].shape, generator=gen)
        m.load_state_dict({**sd, **({"encoder.w": m.state_dict()["encoder.w"]} if "encoder.w" in m.state_dict() else {})})
        t = torch.tensor([10.0, 500.0, 999.0, 3.0])
        l = torch.tensor([0.0, 7.0, 15.0, 2.0])
        out = m(timestep=t, unet_layer=l, input_ids_placeholder_view=None, truncation_idx=None)
        gw = torch.randn(out.word_embedding.shape, generator=gen)
        gb = torch.randn(out.bypass_output.shape, generator=gen)
        ((out.word_embedding * gw).sum() + (out.bypass_output * gb).sum()).backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if k != "encoder.w" and p.grad is not None}
        # oracle
        po = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        wo, bo = R.mapper_forward(po, R.fourier_w([0.03, 2.0]), t, l, 0.4)
        ((wo * gw).sum() + (bo * gb).sum()).backward()
        close(wo, out.word_embedding, 1e-5, f"G2 word D={D}")
        close(bo, out.bypass_output, 1e-5, f"G2 bypass D={D}")
        for k in grads:
            close(po[k].grad, grads[k], 1e-4, f"G2 grad {k} D={D}")
        arrays = {"t": t, "l": l, "word": out.word_embedding, "bypass": out.bypass_output, "gw": gw, "gb": gb,
                  "n_params": np.array(sum(v.numel() for v in sd.values()))}
        if D == 768:  # keep the committed fixture small: full state only once
            arrays.update({"sd." + k: v for k, v in sd.items()})
            arrays.update({"grad." + k: (v if v.numel() < 10000 else v[:, :4]) for k, v in grads.items()})
        else:
            # D=1024: store a seed-regenerable recipe instead of 141k floats x 2
            arrays.update({"sd." + k: v for k, v in sd.items() if "output_layer" not in k})
            arrays["out_w_checksum"] = np.array([sd["output_layer.0.weight"].double().sum().item(),
                                                 sd["output_layer.0.weight"].double().abs().sum().item()])
        save(f"g2_mapper_object_{D}", **arrays)
        print(f"  param count D={D}: {sum(v.numel() for v in sd.values())}")

    # ---------------- G3: view mapper (dtu-12d) with synthetic calibration files --------------
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            cal = os.path.join("data", "dtu", "Calibration", "cal18")
            os.makedirs(cal)
            rng = np.random.RandomState(11)
            mats = rng.randn(64, 3, 4) * np.array([[1e3, 1e3, 1e3, 1e5]])
            for i in range(64):
                np.savetxt(os.path.join(cal, f"pos_{i + 1:03d}.txt"), mats[i])
            from training.dataset import TextualInversionDataset as TID
            toks, lookup = TID.dtu_generate_dset_cam_tokens_params()
            cams = [0, 8, 13, 22, 25, 28]
            view_tokens = [toks[c] for c in cams]
            view_ids = [49410 + i for i in range(len(cams))]
            torch.manual_seed(321)
            mv = NeTIMapper(embedding_type="view", output_dim=768, use_nested_dropout=False,
                            norm_scale=torch.tensor(0.35),
                            pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0, sigma_dtu12=0.5), output_bypass=True,
                            placeholder_view_tokens=list(view_tokens), placeholder_view_token_ids=list(view_ids),
                            arch_view_net=15, arch_view_disable_tl=False, bypass_unconstrained=False,
                            output_bypass_alpha=0.2)
            mv.eval()
            t = torch.tensor([10.0, 500.0, 999.0, 3.0])
            l = torch.tensor([0.0, 7.0, 15.0, 2.0])
            ids = torch.tensor([view_ids[0], view_ids[3], view_ids[5], view_ids[1]])
            out = mv(timestep=t, unet_layer=l, input_ids_placeholder_view=ids, truncation_idx=None)
            params = torch.stack([mv.view_tokenid_2_view_params[i.item()] for i in ids]).float()
            scaled = NeTIMapper.scale_m1_1(params, mv.cam_mins, mv.cam_maxs)
            sdv = {k: v.detach().clone() for k, v in mv.state_dict().items() if k != "encoder.w"}
            wv, bv = R.mapper_forward(sdv, R.fourier_w([0.03, 2.0] + [0.5] * 12), t, l, 0.35, view_params=scaled)
            close(wv, out.word_embedding, 1e-5, "G3 view word")
            close(bv, out.bypass_output, 1e-5, "G3 view bypass")
            # token <-> params round trip
            p0, key0 = TID.dtu_token_to_cam_params(view_tokens[0], cam_idx_as_int=True)
            save("g3_mapper_view", calib=mats.astype(np.float64), cams=np.array(cams), view_ids=np.array(view_ids),
                 tokens=np.array(view_tokens), cam_mins=mv.cam_mins, cam_maxs=mv.cam_maxs, t=t, l=l, ids=ids,
                 params=params, scaled=scaled, word=out.word_embedding, bypass=out.bypass_output,
                 token0_params=p0, token0_key=np.array(key0),
                 **{"sd." + k: v for k, v in sdv.items()})
        finally:
            os.chdir(cwd)

    # ---------------- G4: NeTI text embeddings (tiny CLIP config) ----------------------------
    from transformers import CLIPTextConfig
    tc = CLIPTextConfig(vocab_size=96, hidden_size=32, max_position_embeddings=77, num_hidden_layers=2,
                        num_attention_heads=2, intermediate_size=64)
    torch.manual_seed(9)
    emb = NeTICLIPTextEmbeddings(tc)
    torch.manual_seed(10)
    mo = NeTIMapper(embedding_type="object", output_dim=32, arch_mlp_hidden_dims=64, use_nested_dropout=False,
                    norm_scale=torch.tensor(0.4), pe_sigmas=PESigmas(sigma_t=0.03, sigma_l=2.0), output_bypass=True,
                    arch_view_net=15, arch_view_disable_tl=False, bypass_unconstrained=False, output_bypass_alpha=0.2)
    mo.eval()
    emb.set_mapper({90: mo}, None, device="cpu")
    ids = torch.randint(0, 90, (3, 77), generator=torch.Generator().manual_seed(1))
    pos = [5, 9, 2]
    for b, p in enumerate(pos):
        ids[b, p] = 90
    batch = NeTIBatch(input_ids=ids, input_ids_placeholder_object=torch.tensor([90, 90, 90]),
                      input_ids_placeholder_view=torch.tensor([-1, -1, -1]), timesteps=torch.tensor([3, 700, 42]),
                      unet_layers=torch.tensor([4, 4, 4]))
    with torch.no_grad():
        outs = emb(batch=batch)
    hidden, byp_o, byp_v, unc_o, unc_v, al_o, al_v = outs
    sdo = {k: v.detach().clone() for k, v in mo.state_dict().items() if k != "encoder.w"}
    with torch.no_grad():
        wo, bo = R.mapper_forward(sdo, R.fourier_w([0.03, 2.0]), batch.timesteps, batch.unet_layers, 0.4)
        mine = R.neti_embeddings(emb.token_embedding.weight, emb.position_embedding.weight, ids,
                                 batch.input_ids_placeholder_object, wo)
    close(mine, hidden, 1e-6, "G4 embeddings")
    close(bo, byp_o, 1e-6, "G4 bypass passthrough")
    assert byp_v is None and unc_o is False and unc_v is False and al_o == 0.2 and al_v is None
    save("g4_text_embeddings", token_emb=emb.token_embedding.weight, pos_emb=emb.position_embedding.weight, ids=ids,
         timesteps=batch.timesteps, layers=batch.unet_layers, hidden=hidden, bypass=byp_o,
         **{"sd." + k: v for k, v in sdo.items()})

    # ---------------- G5: XTI attention processor -------------------------------------------
    class Attn:
        def __init__(self, C, Dctx, heads, gen):
            self.heads = heads
            self.to_q = torch.nn.Linear(C, C, bias=False)
            self.to_k = torch.nn.Linear(Dctx, C, bias=False)
            self.to_v = torch.nn.Linear(Dctx, C, bias=False)
            self.to_out = torch.nn.ModuleList([torch.nn.Linear(C, C), torch.nn.Dropout(0.0)])
            self.scale = (C // heads) ** -0.5
            self.cross_attention_norm = False
            for p in list(self.to_q.parameters()) + list(self.to_k.parameters()) + list(self.to_v.parameters()) + \
                    list(self.to_out.parameters()):
                p.data = torch.randn(p.shape, generator=gen) * 0.2

        def prepare_attention_mask(self, mask, n, b):
            return mask

        def head_to_batch_dim(self, t):
            b, n, c = t.shape
            return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n,
                                                                                         c // self.heads)

        def batch_to_head_dim(self, t):
            bh, n, d = t.shape
            b = bh // self.heads
            return t.reshape(b, self.heads, n, d).permute(0, 2, 1, 3).reshape(b, n, d * self.heads)

        def get_attention_scores(self, q, k, mask=None):
            s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1]), q, k.transpose(-1, -2), beta=0,
                              alpha=self.scale)
            return s.softmax(dim=-1)

    gen = torch.Generator().manual_seed(77)
    C_, Dctx, heads = 64, 48, 8
    proc = XTIAttenProc()
    attn_self = Attn(C_, C_, heads, gen)
    attn_cross = Attn(C_, Dctx, heads, gen)
    hs = torch.randn(2, 10, C_, generator=gen)
    ctx = {"this_idx": 14}
    for i in range(16):
        ctx[f"CONTEXT_TENSOR_{i}"] = torch.randn(2, 7, Dctx, generator=gen)
        ctx[f"CONTEXT_TENSOR_BYPASS_{i}"] = torch.randn(2, 7, Dctx, generator=gen)
    ctx_o = dict(ctx)
    with torch.no_grad():
        y_self = proc(attn_self, hs, None)
        y_tensor = proc(attn_cross, hs, ctx["CONTEXT_TENSOR_3"])
        seq = []
        ys = []
        for _ in range(3):
            ys.append(proc(attn_cross, hs, ctx))
            seq.append(ctx["this_idx"])

        def mine(a, h, e):
            return R.xti_attention(a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_out[0].weight, a.to_out[0].bias,
                                   heads, h, e)
        close(mine(attn_self, hs, None), y_self, 1e-5, "G5 self")
        close(mine(attn_cross, hs, ctx_o["CONTEXT_TENSOR_3"]), y_tensor, 1e-5, "G5 tensor ctx")
        seq_o = []
        for k in range(3):
            close(mine(attn_cross, hs, ctx_o), ys[k], 1e-5, f"G5 dict ctx call {k}")
            seq_o.append(ctx_o["this_idx"])
    assert seq == seq_o == [15, 0, 1], (seq, seq_o)
    save("g5_xti_attention", hs=hs, wq=attn_cross.to_q.weight, wk=attn_cross.to_k.weight, wv=attn_cross.to_v.weight,
         wo=attn_cross.to_out[0].weight, bo=attn_cross.to_out[0].bias, ctx14=ctx["CONTEXT_TENSOR_14"],
         ctxb14=ctx["CONTEXT_TENSOR_BYPASS_14"], ctx15=ctx["CONTEXT_TENSOR_15"], ctxb15=ctx["CONTEXT_TENSOR_BYPASS_15"],
         ctx0=ctx["CONTEXT_TENSOR_0"], ctxb0=ctx["CONTEXT_TENSOR_BYPASS_0"], y0=ys[0], y1=ys[1], y2=ys[2],
         seq=np.array(seq), sq=attn_self.to_q.weight, sk=attn_self.to_k.weight, sv=attn_self.to_v.weight,
         so=attn_self.to_out[0].weight, sbo=attn_self.to_out[0].bias, y_self=y_self)

    # ---------------- G5b: the same processor at head dims the HIP kernels implement (40, 64), with gradients ---
    for tag, C2, heads2, Dctx2 in (("d40", 320, 8, 128), ("d64", 128, 2, 64)):
        gen = torch.Generator().manual_seed(88)
        r16 = lambda x: x.half().float()  # f16-representable values: the GPU test feeds the very same numbers
        proc = XTIAttenProc()
        a_self, a_cross = Attn(C2, C2, heads2, gen), Attn(C2, Dctx2, heads2, gen)
        for a in (a_self, a_cross):
            for prm in list(a.to_q.parameters()) + list(a.to_k.parameters()) + list(a.to_v.parameters()) + \
                    list(a.to_out.parameters()):
                prm.data = r16(prm.data * (0.5 / 0.2) / (prm.shape[-1] ** 0.5))
        Bn, Nq, Nk = 2, 24, 7
        hs = r16(torch.randn(Bn, Nq, C2, generator=gen)).requires_grad_(True)
        cd = {"this_idx": 15}
        for i in (15, 0):
            cd[f"CONTEXT_TENSOR_{i}"] = r16(torch.randn(Bn, Nk, Dctx2, generator=gen)).requires_grad_(True)
            cd[f"CONTEXT_TENSOR_BYPASS_{i}"] = r16(torch.randn(Bn, Nk, Dctx2, generator=gen)).requires_grad_(True)
        gy = r16(torch.randn(Bn, Nq, C2, generator=gen))
        y_self = proc(a_self, hs, None)
        (g_self,) = torch.autograd.grad((y_self * gy).sum(), hs)
        y_t = proc(a_cross, hs, cd["CONTEXT_TENSOR_0"])
        g_t = torch.autograd.grad((y_t * gy).sum(), [hs, cd["CONTEXT_TENSOR_0"]])
        ys, gs, seq = [], [], []
        for _ in range(2):
            i = cd["this_idx"]
            y = proc(a_cross, hs, cd)
            seq.append(cd["this_idx"])
            ys.append(y)
            gs.append(torch.autograd.grad((y * gy).sum(), [hs, cd[f"CONTEXT_TENSOR_{i}"],
                                                           cd[f"CONTEXT_TENSOR_BYPASS_{i}"]]))
        assert seq == [0, 1]
        with torch.no_grad():
            mine = R.xti_attention(a_cross.to_q.weight, a_cross.to_k.weight, a_cross.to_v.weight, a_cross.to_out[0].weight,
                                   a_cross.to_out[0].bias, heads2, hs, {"this_idx": 15, **{k: v for k, v in cd.items()
                                                                                        if k != "this_idx"}})
        close(mine, ys[0], 1e-5, f"G5b {tag} dict ctx")
        h16 = lambda x: x.detach().half()
        save(f"g5b_xti_attention_{tag}", heads=np.array(heads2), hs=h16(hs), gy=h16(gy),
             wq=h16(a_cross.to_q.weight), wk=h16(a_cross.to_k.weight), wv=h16(a_cross.to_v.weight),
             wo=h16(a_cross.to_out[0].weight), bo=a_cross.to_out[0].bias.detach(),
             sq=h16(a_self.to_q.weight), sk=h16(a_self.to_k.weight), sv=h16(a_self.to_v.weight),
             so=h16(a_self.to_out[0].weight), sbo=a_self.to_out[0].bias.detach(),
             ctx15=h16(cd["CONTEXT_TENSOR_15"]), ctxb15=h16(cd["CONTEXT_TENSOR_BYPASS_15"]),
             ctx0=h16(cd["CONTEXT_TENSOR_0"]), ctxb0=h16(cd["CONTEXT_TENSOR_BYPASS_0"]),
             y_self=y_self.detach(), g_self=g_self, y_tensor=y_t.detach(), g_tensor_hs=g_t[0], g_tensor_ctx=g_t[1],
             y0=ys[0].detach(), y1=ys[1].detach(), g0_hs=gs[0][0], g0_ctx=gs[0][1], g0_ctxb=gs[0][2],
             g1_hs=gs[1][0], g1_ctx=gs[1][1], g1_ctxb=gs[1][2], seq=np.array(seq))

    # ---------------- G6: third-party CLIP text stack (transformers, random tiny weights) -----
    from transformers import CLIPTextModel
    from view_neti_amd import sd_config as sc
    for act in ("quick_gelu", "gelu"):
        tc = CLIPTextConfig(vocab_size=96, hidden_size=64, max_position_embeddings=77, num_hidden_layers=2,
                            num_attention_heads=4, intermediate_size=128, hidden_act=act, eos_token_id=95)
        torch.manual_seed(4)
        tm = CLIPTextModel(tc).eval()
        sdt = {k: v.detach().clone() for k, v in tm.state_dict().items()}
        if not any(k.startswith("text_model.") for k in sdt):
            sdt = {"text_model." + k: v for k, v in sdt.items()}
        sdt = {k: v for k, v in sdt.items() if "position_ids" not in k}
        ids = torch.randint(0, 96, (2, 77), generator=torch.Generator().manual_seed(2))
        with torch.no_grad():
            ref = tm(input_ids=ids).last_hidden_state
            my_cfg = sc.CLIPTextConfig(vocab_size=96, hidden_size=64, num_layers=2, num_heads=4, intermediate_size=128,
                                       act=act)
            mine_last, _ = R.neti_text_encoder(sdt, my_cfg, ids, None, None, None)
        close(mine_last, ref, 2e-5, f"G6 CLIP text stack ({act})")
        save(f"g6_clip_tiny_{act}", ids=ids, last_hidden=ref, **{"sd." + k: v for k, v in sdt.items()})

    # ---------------- G8: small helpers --------------------------------------------------------
    xs = torch.tensor([[-3.0, 0.5, 2.0], [1.0, 1.0, 1.0]])
    s1 = NeTIMapper.scale_m1_1(xs, torch.tensor([-3.0, 0.0, 1.0]), torch.tensor([1.0, 1.0, 3.0]))
    nums = [0, 3.0, 1.25, -2.5, 1234.56789, 0.0001]
    strs2 = [num_to_string(n) for n in nums]
    strs4 = [num_to_string(n, tol=4) for n in nums]
    back = [string_to_num(s) for s in strs4]
    save("g8_helpers", xs=xs, scaled=s1, nums=np.array(nums), strs2=np.array(strs2), strs4=np.array(strs4),
         back=np.array(back))

    # ---------------- G6 (config post-init dumps) and G7 (dataset statics) ---------------------------------
    g6_config_dumps()
    g7_dataset_statics()

    # ---------------- G9: the legacy (dataclass-default) object mapper ------------------------------------------
    g9_legacy_mapper(R, NeTIMapper, PESigmas)

    # ---------------- f2: checkpoints in the reference's layout holding the reference's own objects ---------
    f2_reference_checkpoints(R, NeTIMapper, PESigmas)

    # ---------------- G7 (text encoder): the real NeTICLIPTextModel incl. the bypass injection ------------------
    g_text_encoder_bypass(R, NeTIMapper, NeTIBatch, PESigmas)

    # ---------------- G10 / G11: the inference path's two reference-owned files, run for real ------------------
    g10_sd_pipeline_call(R)
    g11_prompt_manager(R, NeTIMapper, NeTIBatch, PESigmas)
    print("all fixtures written and cross-checked against oracle/sd_ref.py")


if __name__ == "__main__":
    main()
