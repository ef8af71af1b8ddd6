"""Text engine (batched 16-pass CLIP + fused NeTI mapper) vs the CPU oracle: both contexts and
the mapper parameter gradients (the only weight gradients of the train step)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


# (view mapper?, object bypass unconstrained?, view bypass unconstrained?, nested dropout?)
VARIANTS = [(False, False, False, False), (True, False, False, False), (False, True, False, False),
            (True, True, True, False), (True, False, True, True), (False, False, False, True)]


@pytest.mark.parametrize("with_view,unc_obj,unc_view,dropout", VARIANTS)
def test_text_engine_tiny(with_view, unc_obj, unc_view, dropout):
    from oracle import sd_ref as R
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.text import MapperState, TextEngine, flatten_mapper_state
    from view_neti_amd.mapper import fourier_frequencies
    dev = "cuda"
    cfg = sc.tiny().clip
    D, L, nl, B = cfg.hidden_size, cfg.max_positions, 16, 2
    w = synth.clip_weights(cfg)
    # matmul operands are fp16 on the GPU: round those weights once for both sides
    wr = {k: (v.half().float() if (k.endswith("weight") and v.dim() == 2 and "embedding" not in k) else v)
          for k, v in w.items()}
    ph_obj, ph_view = cfg.vocab_size - 3, cfg.vocab_size - 4
    ids = synth.input_ids(B, ph_obj, cfg.vocab_size, L, view_placeholder_id=ph_view if with_view else None)
    ids[1] = torch.roll(ids[1], 3)  # different placeholder positions per sample
    t = torch.tensor([17, 803])
    gen = torch.Generator().manual_seed(3)

    def rand_mapper(nfeat_sigmas):
        sd = {"net.0.weight": torch.randn(64, 64, generator=gen) * 0.15, "net.0.bias": torch.randn(64, generator=gen) * 0.1,
              "net.1.weight": 1 + 0.1 * torch.randn(64, generator=gen), "net.1.bias": 0.1 * torch.randn(64, generator=gen),
              "net.3.weight": torch.randn(64, 64, generator=gen) * 0.15, "net.3.bias": torch.randn(64, generator=gen) * 0.1,
              "net.4.weight": 1 + 0.1 * torch.randn(64, generator=gen), "net.4.bias": 0.1 * torch.randn(64, generator=gen),
              "output_layer.0.weight": torch.randn(2 * D, 64, generator=gen) * 0.15,
              "output_layer.0.bias": torch.randn(2 * D, generator=gen) * 0.1}
        return sd, fourier_frequencies(nfeat_sigmas, 64, 0, preserve_rng=True)

    sdo, w_enc_o = rand_mapper([0.03, 2.0])
    ctx_k = torch.zeros(nl, B * L, D, dtype=torch.float16, device=dev)
    ctx_v = torch.zeros_like(ctx_k)
    dk = (synth.gaussian((nl, B * L, D), 11) * 0.5).half()
    dv = (synth.gaussian((nl, B * L, D), 12) * 0.5).half()
    ts = t.to(dev)
    po = flatten_mapper_state(sdo).to(dev)
    go = torch.zeros_like(po)
    pdrop = 0.5 if dropout else 0.0
    rng_state = torch.tensor([77, 5], dtype=torch.int32, device=dev)
    mo = MapperState(po, w_enc_o.to(dev), 0.4, 0.2, unconstrained=unc_obj, nested_dropout_prob=pdrop)
    kw = {}
    view = None
    if with_view:
        sdv, w_enc_v = rand_mapper([0.03, 2.0] + [0.5] * 12)
        vparams = torch.rand(B, 12, generator=gen) * 2 - 1
        pv = flatten_mapper_state(sdv).to(dev)
        gv = torch.zeros_like(pv)
        kw = dict(mapper_view=MapperState(pv, w_enc_v.to(dev), 0.35, 0.3, unconstrained=unc_view,
                                          nested_dropout_prob=pdrop), grads_view=gv)
    eng = TextEngine(cfg, wr, nl, B, ts, ctx_k, ctx_v, dk.to(dev), dv.to(dev), mo, go, rng_state=rng_state, **kw)
    eng.set_batch(ids, torch.full((B,), ph_obj), torch.full((B,), ph_view) if with_view else None,
                  vparams if with_view else None)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    # ---- nested-dropout masks drawn on the device: structure check, then hand them to the oracle ----
    masks_o = masks_v = None
    if dropout:
        def check(mask):
            m = mask.cpu().view(nl, B, -1)
            assert set(m.unique().tolist()) <= {0.0, 1.0}
            kept = m.sum(-1)
            # a prefix of ones per row (hidden[idx:] = 0), whole layers either fire or not
            assert bool((m[..., :-1] >= m[..., 1:]).all())
            fired = (kept < m.shape[-1]).any(1)
            assert 0 < int(fired.sum()) < nl, "p=0.5 over 16 layers: some but not all mapper calls drop"
            return m
        masks_o = check(eng.hidden_mask_obj)
        if with_view:
            masks_v = check(eng.hidden_mask_view)
            assert not torch.equal(masks_o, masks_v)
    # ---- oracle ----
    p_o = {k: v.clone().requires_grad_(True) for k, v in sdo.items()}
    if with_view:
        p_v = {k: v.clone().requires_grad_(True) for k, v in sdv.items()}
        view = dict(p=p_v, w_enc=w_enc_v, norm_scale=0.35, placeholder=torch.full((B,), ph_view), params=vparams,
                    alpha=0.3, unconstrained=unc_view, hidden_masks=masks_v)
    hs = R.text_conditioning(wr, cfg, p_o, w_enc_o, 0.4, ids, torch.full((B,), ph_obj), t, alpha=0.2,
                             unconstrained=unc_obj, n_layers=nl, view=view, hidden_masks=masks_o)
    rk = torch.stack([hs[f"CONTEXT_TENSOR_{i}"] for i in range(nl)]).reshape(nl, B * L, D)
    rv = torch.stack([hs[f"CONTEXT_TENSOR_BYPASS_{i}"] for i in range(nl)]).reshape(nl, B * L, D)
    ek, ev = _rel(ctx_k, rk.detach()), _rel(ctx_v, rv.detach())
    print(f"[text view={with_view} unc={unc_obj}/{unc_view} drop={dropout}] ctx_k rel {ek:.3e}  ctx_v rel {ev:.3e}")
    assert ek < 5e-3 and ev < 5e-3
    ((rk * dk.float()).sum() + (rv * dv.float()).sum()).backward()
    ref_g = flatten_mapper_state({k: v.grad for k, v in p_o.items()})
    eg = _rel(go, ref_g)
    cos = torch.nn.functional.cosine_similarity(go.float().cpu(), ref_g, dim=0).item()
    print(f"[text view={with_view}] object-mapper grad rel {eg:.3e} cos {cos:.6f} |g| {ref_g.norm():.3e}")
    assert math.isfinite(eg) and eg < 3e-2 and cos > 0.999
    if with_view:
        ref_gv = flatten_mapper_state({k: v.grad for k, v in p_v.items()})
        egv = _rel(gv, ref_gv)
        cosv = torch.nn.functional.cosine_similarity(gv.float().cpu(), ref_gv, dim=0).item()
        print(f"[text view] view-mapper grad rel {egv:.3e} cos {cosv:.6f}")
        assert egv < 3e-2 and cosv > 0.999


@pytest.mark.parametrize("dropout", [False, True])
def test_text_engine_legacy_object_mapper(dropout):
    """SURVEY a5': the reference's dataclass-default object mapper (arch_view_net 0 — NeTIPositionalEncoding of the raw
    (t, l), anchor-initialised trainable input_layer 2048 -> 160, MLP h = 128; models/positional_encoding.py:10-51,
    models/neti_mapper.py:155-163) on the HIP text path: both contexts and ALL parameter gradients (input_layer included)
    against the oracle, whose legacy restatement G9 pins to the real module."""
    from oracle import sd_ref as R
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.text import MapperState, TextEngine, flatten_mapper_state, unflatten_mapper_state
    dev = "cuda"
    cfg = sc.tiny().clip
    D, L, nl, B, E, h, P2 = cfg.hidden_size, cfg.max_positions, 16, 2, 160, 128, 2048
    w = synth.clip_weights(cfg)
    wr = {k: (v.half().float() if (k.endswith("weight") and v.dim() == 2 and "embedding" not in k) else v)
          for k, v in w.items()}
    ph_obj = cfg.vocab_size - 3
    ids = synth.input_ids(B, ph_obj, cfg.vocab_size, L)
    ids[1] = torch.roll(ids[1], 3)
    t = torch.tensor([17, 803])
    gen = torch.Generator().manual_seed(5)
    w_pe = torch.randn(1024, 2, generator=gen) * torch.tensor([0.03, 2.0])
    rn = lambda *s, sc_=0.1: torch.randn(*s, generator=gen) * sc_
    sd = {"input_layer.weight": R.neti_pe_init_layer(w_pe) + rn(E, P2, sc_=0.02), "input_layer.bias": rn(E),
          "net.0.weight": rn(h, E, sc_=0.15), "net.0.bias": rn(h), "net.1.weight": 1 + rn(h), "net.1.bias": rn(h),
          "net.3.weight": rn(h, h, sc_=0.1), "net.3.bias": rn(h), "net.4.weight": 1 + rn(h), "net.4.bias": rn(h),
          "output_layer.0.weight": rn(2 * D, h, sc_=0.15), "output_layer.0.bias": rn(2 * D)}
    flat = flatten_mapper_state(sd)
    back = unflatten_mapper_state(flat, E, h, 2 * D, P2)
    assert all(torch.equal(back[k], sd[k]) for k in sd) and flat.numel() == sum(v.numel() for v in sd.values())
    ctx_k = torch.zeros(nl, B * L, D, dtype=torch.float16, device=dev)
    ctx_v = torch.zeros_like(ctx_k)
    dk = (synth.gaussian((nl, B * L, D), 11) * 0.5).half()
    dv = (synth.gaussian((nl, B * L, D), 12) * 0.5).half()
    po = flat.to(dev)
    go = torch.full_like(po, 7.0)  # must be overwritten, not accumulated into
    rng_state = torch.tensor([77, 5], dtype=torch.int32, device=dev)
    mo = MapperState(po, None, 0.4, 0.2, hidden=h, enc_dim=E, nested_dropout_prob=0.5 if dropout else 0.0,
                     legacy_w_pe=w_pe.to(dev))
    eng = TextEngine(cfg, wr, nl, B, t.to(dev), ctx_k, ctx_v, dk.to(dev), dv.to(dev), mo, go, rng_state=rng_state)
    eng.set_batch(ids, torch.full((B,), ph_obj))
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    masks = eng.hidden_mask_obj.cpu().view(nl, B, -1) if dropout else None
    p_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hs = R.text_conditioning(wr, cfg, p_o, w_pe, 0.4, ids, torch.full((B,), ph_obj), t, alpha=0.2, n_layers=nl,
                             hidden_masks=masks)
    rk = torch.stack([hs[f"CONTEXT_TENSOR_{i}"] for i in range(nl)]).reshape(nl, B * L, D)
    rv = torch.stack([hs[f"CONTEXT_TENSOR_BYPASS_{i}"] for i in range(nl)]).reshape(nl, B * L, D)
    ek, ev = _rel(ctx_k, rk.detach()), _rel(ctx_v, rv.detach())
    ((rk * dk.float()).sum() + (rv * dv.float()).sum()).backward()
    ref_g = flatten_mapper_state({k: v.grad for k, v in p_o.items()})
    eg = _rel(go, ref_g)
    cos = torch.nn.functional.cosine_similarity(go.float().cpu(), ref_g, dim=0).item()
    n_in = E * P2 + E
    eg_in = _rel(go[-n_in:], ref_g[-n_in:])
    print(f"[text legacy mapper drop={dropout}] ctx_k rel {ek:.3e} ctx_v rel {ev:.3e}; grad rel {eg:.3e} cos {cos:.6f}; "
          f"input_layer grad rel {eg_in:.3e} |g_in| {ref_g[-n_in:].norm():.3e}")
    assert ek < 5e-3 and ev < 5e-3
    assert math.isfinite(eg) and eg < 3e-2 and cos > 0.999 and eg_in < 3e-2
    # accumulate mode adds on top
    eng.accumulate_grads = True
    eng.backward()
    torch.cuda.synchronize()
    assert _rel(go, 2 * ref_g) < 3e-2


def test_module_call_protocol_text_encoder():
    """Seam B: `text_encoder(batch=NeTIBatch)` served by the HIP engine — the reference's calling pattern
    (one call per UNet layer, `[0]` of both returned outputs; training/coach.py:289-305) against the oracle's
    restatement of NeTICLIPTextTransformer.forward, which g7_text_encoder_bypass pins to the real module."""
    from oracle import sd_ref as R
    from utils.types import NeTIBatch
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.compat.neti_clip_text_encoder import HipNeTICLIPTextModel
    from view_neti_amd.compat.neti_modules import NeTIMapper
    cfg = sc.tiny().clip
    D, L, B = cfg.hidden_size, cfg.max_positions, 2
    w = synth.clip_weights(cfg)
    wr = {k: (v.half().float() if (k.endswith("weight") and v.dim() == 2 and "embedding" not in k) else v)
          for k, v in w.items()}
    ph = cfg.vocab_size - 3
    torch.manual_seed(4)
    mapper = NeTIMapper("object", D, 64, 0.4, output_bypass_alpha=0.2, placeholder_object_token="<obj>")
    with torch.no_grad():
        for prm in mapper.parameters():
            prm.add_(0.1 * torch.randn(prm.shape))
    enc = HipNeTICLIPTextModel(cfg, wr)
    enc.text_model.embeddings.set_mapper({ph: mapper}, None)
    assert enc.get_input_embeddings().weight.shape == (cfg.vocab_size, D)
    ids = synth.input_ids(B, ph, cfg.vocab_size, L)
    t = torch.tensor([17, 803])
    sd = {k: v.detach() for k, v in mapper.mapper_state().items()}
    for layer in (0, 7, 15):
        batch = NeTIBatch(input_ids=ids, input_ids_placeholder_object=torch.full((B,), ph),
                          input_ids_placeholder_view=torch.full((B,), -1), timesteps=t,
                          unet_layers=torch.full((B,), layer))
        out, out_b = enc(batch=batch)
        word, byp = R.mapper_forward(sd, mapper.encoder.w, t, torch.full((B,), float(layer)), 0.4)
        ref, ref_b = R.neti_text_encoder(wr, cfg, ids, torch.full((B,), ph), word, byp, False, 0.2)
        e0, e1 = _rel(out[0], ref), _rel(out_b[0], ref_b)
        print(f"[module-call protocol layer {layer}] last_hidden_state rel {e0:.2e}, with bypass rel {e1:.2e}")
        assert out[0].shape == (B, L, D) and e0 < 5e-3 and e1 < 5e-3
        assert torch.equal(out.last_hidden_state, out[0]) and out.pooler_output.shape == (B, D)
    # an in-place update of the mapper (optimizer step, load_state_dict) must be seen: the adapter may neither serve its
    # cached result nor run the engine on its stale copy of the parameters
    with torch.no_grad():
        for prm in mapper.parameters():
            prm.add_(0.3 * torch.randn(prm.shape))
    sd2 = {k: v.detach() for k, v in mapper.mapper_state().items()}
    out2, out2_b = enc(batch=batch)
    word2, byp2 = R.mapper_forward(sd2, mapper.encoder.w, t, torch.full((B,), float(layer)), 0.4)
    ref2, ref2_b = R.neti_text_encoder(wr, cfg, ids, torch.full((B,), ph), word2, byp2, False, 0.2)
    moved = _rel(ref2, ref)
    print(f"[module-call protocol] after an in-place mapper update: oracle moved by {moved:.2e}, adapter vs new oracle "
          f"{_rel(out2[0], ref2):.2e} / {_rel(out2_b[0], ref2_b):.2e}")
    assert moved > 2e-2 and _rel(out2[0], ref2) < 5e-3 and _rel(out2_b[0], ref2_b) < 5e-3
    # the plain input_ids= path (negative prompt, sd_pipeline_call.py:35-39): no bypass variant
    plain, none = enc(input_ids=ids)
    assert none is None and _rel(plain[0], R.clip_plain(wr, cfg, ids)) < 5e-3


@pytest.mark.parametrize("with_view,byp_obj,byp_view", [(False, False, True), (True, False, True), (True, True, False),
                                                        (True, False, False)])
def test_text_engine_mappers_without_textual_bypass(with_view, byp_obj, byp_view):
    """`output_bypass_{object,view} = False` (models/neti_mapper.py:79-81,419-424; net_clip_text_embedding.py:88-90): such a
    mapper emits the word embedding only (output layer D wide); where no mapper emits a bypass the value context is the key
    context (`CONTEXT_TENSOR_BYPASS_i` absent, xti_attention_processor.py:19-20,39-42)."""
    from oracle import sd_ref as R
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.text import MapperState, TextEngine, flatten_mapper_state
    from view_neti_amd.mapper import fourier_frequencies
    dev = "cuda"
    cfg = sc.tiny().clip
    D, L, nl, B = cfg.hidden_size, cfg.max_positions, 16, 2
    w = synth.clip_weights(cfg)
    wr = {k: (v.half().float() if (k.endswith("weight") and v.dim() == 2 and "embedding" not in k) else v)
          for k, v in w.items()}
    ph_obj, ph_view = cfg.vocab_size - 3, cfg.vocab_size - 4
    ids = synth.input_ids(B, ph_obj, cfg.vocab_size, L, view_placeholder_id=ph_view if with_view else None)
    t = torch.tensor([17, 803])
    gen = torch.Generator().manual_seed(3)
    rn = lambda *s, sc_=0.1: torch.randn(*s, generator=gen) * sc_

    def rand_mapper(sigmas, bypass):
        od = 2 * D if bypass else D
        sd = {"net.0.weight": rn(64, 64, sc_=0.15), "net.0.bias": rn(64), "net.1.weight": 1 + rn(64), "net.1.bias": rn(64),
              "net.3.weight": rn(64, 64, sc_=0.15), "net.3.bias": rn(64), "net.4.weight": 1 + rn(64), "net.4.bias": rn(64),
              "output_layer.0.weight": rn(od, 64, sc_=0.15), "output_layer.0.bias": rn(od)}
        return sd, fourier_frequencies(sigmas, 64, 0, preserve_rng=True)

    sdo, w_o = rand_mapper([0.03, 2.0], byp_obj)
    ctx_k = torch.zeros(nl, B * L, D, dtype=torch.float16, device=dev)
    ctx_v = torch.zeros_like(ctx_k)
    dk = (synth.gaussian((nl, B * L, D), 11) * 0.5).half()
    dv = (synth.gaussian((nl, B * L, D), 12) * 0.5).half()
    po = flatten_mapper_state(sdo).to(dev)
    go = torch.zeros_like(po)
    mo = MapperState(po, w_o.to(dev), 0.4, 0.2, output_bypass=byp_obj)
    kw, view = {}, None
    if with_view:
        sdv, w_v = rand_mapper([0.03, 2.0] + [0.5] * 12, byp_view)
        vparams = torch.rand(B, 12, generator=gen) * 2 - 1
        pv = flatten_mapper_state(sdv).to(dev)
        gv = torch.zeros_like(pv)
        kw = dict(mapper_view=MapperState(pv, w_v.to(dev), 0.35, 0.3, output_bypass=byp_view), grads_view=gv)
    eng = TextEngine(cfg, wr, nl, B, t.to(dev), ctx_k, ctx_v, dk.to(dev), dv.to(dev), mo, go, **kw)
    eng.set_batch(ids, torch.full((B,), ph_obj), torch.full((B,), ph_view) if with_view else None,
                  vparams if with_view else None)
    eng.forward()
    eng.backward()
    torch.cuda.synchronize()
    p_o = {k: v.clone().requires_grad_(True) for k, v in sdo.items()}
    if with_view:
        p_v = {k: v.clone().requires_grad_(True) for k, v in sdv.items()}
        view = dict(p=p_v, w_enc=w_v, norm_scale=0.35, placeholder=torch.full((B,), ph_view), params=vparams, alpha=0.3,
                    output_bypass=byp_view)
    hs = R.text_conditioning(wr, cfg, p_o, w_o, 0.4, ids, torch.full((B,), ph_obj), t, alpha=0.2, n_layers=nl, view=view,
                             output_bypass=byp_obj)
    any_bypass = byp_obj or (with_view and byp_view)
    assert ("CONTEXT_TENSOR_BYPASS_0" in hs) == any_bypass
    rk = torch.stack([hs[f"CONTEXT_TENSOR_{i}"] for i in range(nl)]).reshape(nl, B * L, D)
    rv = torch.stack([hs.get(f"CONTEXT_TENSOR_BYPASS_{i}", hs[f"CONTEXT_TENSOR_{i}"]) for i in range(nl)]).reshape(nl, B * L, D)
    ek, ev = _rel(ctx_k, rk.detach()), _rel(ctx_v, rv.detach())
    assert ek < 5e-3 and ev < 5e-3
    if not any_bypass:
        assert torch.equal(ctx_k, ctx_v)
    ((rk * dk.float()).sum() + (rv * dv.float()).sum()).backward()
    ref_g = flatten_mapper_state({k: v.grad for k, v in p_o.items()})
    eg = _rel(go, ref_g)
    print(f"[text no-bypass view={with_view} obj_bypass={byp_obj} view_bypass={byp_view}] ctx_k {ek:.2e} ctx_v {ev:.2e} "
          f"object grad rel {eg:.2e} ({ref_g.numel()} params)")
    assert eg < 3e-2
    if with_view:
        ref_gv = flatten_mapper_state({k: v.grad for k, v in p_v.items()})
        assert _rel(gv, ref_gv) < 3e-2
