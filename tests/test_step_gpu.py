"""Whole train step (VAE -> noise -> 16x text -> UNet -> MSE -> backward -> AdamW) on the GPU vs
the CPU oracle.  Parity bar (BASELINE.json north_star / SURVEY §8d): predicted-noise MSE within
1e-3 relative, mapper-gradient cosine >= 0.999."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def build(cfg_name, B, H, W, with_view=False, **kw):
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.step import TrainStepEngine
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    cfg = sc.CONFIGS[cfg_name]()
    dev = "cuda"
    gpu = not cfg_name.startswith("tiny")
    uw = synth.unet_weights(cfg.unet, device=dev if gpu else "cpu")
    vw = synth.vae_weights(cfg.vae, device=dev if gpu else "cpu")
    cw = synth.clip_weights(cfg.clip, device=dev if gpu else "cpu")
    D = cfg.clip.hidden_size
    torch.manual_seed(0)
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0)
    sd = init_mapper_state(64, 64, D)
    gen = torch.Generator().manual_seed(1)
    sd = {k: v + 0.05 * torch.randn(v.shape, generator=gen) for k, v in sd.items()}
    extra = {}
    if with_view:
        w_enc_v = fourier_frequencies([0.03, 2.0] + [0.5] * 12, 64, 0)
        sdv = init_mapper_state(64, 64, D)
        sdv = {k: v + 0.05 * torch.randn(v.shape, generator=gen) for k, v in sdv.items()}
        extra = dict(mapper_view=sdv, w_enc_view=w_enc_v, norm_scale_view=0.35, alpha_view=0.3)
    eng = TrainStepEngine(cfg, uw, vw, cw, B, H, W, sd, w_enc, 0.4, 0.2, **extra, **kw)
    return cfg, eng, (uw, vw, cw), sd, w_enc, extra


# config 2 (mode 0, SD-1.5 family) / config 3 (mode 2: object + view mappers) / SD-2.1 family switches
# (linear projections, GELU CLIP, v-prediction; non-square DTU-like frame) / unconstrained bypass
@pytest.mark.parametrize("cfg_name,with_view,H,W,unc", [("tiny", False, 64, 64, False), ("tiny", True, 64, 64, False),
                                                        ("tiny21", True, 64, 128, False),
                                                        ("tiny21", True, 64, 64, True)])
def test_train_step_tiny_matches_oracle(cfg_name, with_view, H, W, unc):
    from oracle import sd_ref as R
    from view_neti_amd import synth
    from view_neti_amd.engine.text import flatten_mapper_state
    B = 2
    lr = 4e-3
    cfg, eng, (uw, vw, cw), sd, w_enc, extra = build(cfg_name, B, H, W, with_view, device_rng=False, lr=lr,
                                                     unconstrained_object=unc, unconstrained_view=unc)
    ph = cfg.clip.vocab_size - 3
    phv = cfg.clip.vocab_size - 4
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv if with_view else None)
    px = synth.pixel_values(B, H, W)
    t = synth.timesteps(B)
    eps = synth.gaussian((B, 4, H // 8, W // 8), 3)
    noise = synth.gaussian((B, 4, H // 8, W // 8), 4)
    vparams = synth.gaussian((B, 12), 9).clamp(-1, 1) if with_view else None
    eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv) if with_view else None, vparams)
    eng.set_noise(eps, noise, t)
    p0 = eng.params.clone()
    eng.forward_backward()
    torch.cuda.synchronize()
    loss_gpu = eng.loss()
    grads_gpu = (eng.grads / eng.scaler[0]).float().cpu()
    # ---- oracle: same weights (fp16-rounded where the GPU holds fp16), fp32 arithmetic ----
    r16 = lambda d: {k: (v.half().float() if v.dim() >= 2 and "embedding" not in k else v) for k, v in d.items()}
    p_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    view = None
    if with_view:
        p_v = {k: v.clone().requires_grad_(True) for k, v in extra["mapper_view"].items()}
        view = dict(p=p_v, w_enc=extra["w_enc_view"], norm_scale=0.35, placeholder=torch.full((B,), phv),
                    params=vparams, alpha=0.3, unconstrained=unc)
    loss, aux = R.train_step_loss(cfg, r16(uw), r16(vw), r16(cw), p_o, w_enc, 0.4, px, ids, torch.full((B,), ph), t,
                                  eps, noise, alpha=0.2, unconstrained=unc, view=view)
    loss.backward()
    ref = [flatten_mapper_state({k: v.grad for k, v in p_o.items()})]
    if with_view:
        ref.append(flatten_mapper_state({k: v.grad for k, v in p_v.items()}))
    ref_g = torch.cat(ref)
    rel = abs(loss_gpu - loss.item()) / loss.item()
    cos = torch.nn.functional.cosine_similarity(grads_gpu, ref_g, dim=0).item()
    gerr = ((grads_gpu - ref_g).norm() / ref_g.norm()).item()
    lat = ((eng.latents.cpu() - aux["latents"]).norm() / aux["latents"].norm()).item()
    print(f"[step {cfg_name} {H}x{W} view={with_view} unc={unc}] loss gpu {loss_gpu:.6f} oracle {loss.item():.6f} rel {rel:.2e}; "
          f"latents rel {lat:.2e}; grad cos {cos:.6f} rel {gerr:.2e} |g| {ref_g.norm():.3e}")
    assert rel < 1e-3, "predicted-noise MSE must match the oracle to 1e-3 relative"
    assert cos > 0.999 and gerr < 5e-2
    # ---- optimizer: one AdamW step with GradScaler unscale ----
    eng.optimizer_step()
    torch.cuda.synchronize()
    # (fed with the GPU's own unscaled gradients: at step 1 Adam moves every weight by ~lr*sign(g),
    #  so sign flips of near-zero gradients would otherwise dominate the comparison)
    p1, _, _ = R.adamw_step(p0.cpu(), grads_gpu, torch.zeros_like(ref_g), torch.zeros_like(ref_g), 1, lr)
    perr = ((eng.params.cpu() - p1).abs().max() / lr).item()
    print(f"[step tiny] AdamW max |dp|/lr deviation {perr:.3e}; opt_step {eng.opt_step.item()} scale {eng.scaler[0].item()}")
    assert perr < 0.05 and eng.opt_step.item() == 1


def test_overflow_skips_step():
    """GradScaler semantics (accelerate's fp16 scaler behind training/coach.py:211-218): a loss scale that overflows the f16
    gradients must be DETECTED — the inf has to survive element-wise through every normalisation whose statistics it
    wrecks (csrc/common.h vn_fx_encode) down to the gradient bucket — and the step skipped: parameters, moments and the
    optimizer step count unchanged, the scale halved, found_inf cleared; the next step at the lower scale trains."""
    from view_neti_amd import synth
    B, H, W = 2, 64, 64
    cfg, eng, _, _, _, _ = build("tiny", B, H, W, device_rng=False, lr=1e-3, loss_scale=2.0 ** 40)
    ph = cfg.clip.vocab_size - 3
    eng.set_batch(synth.pixel_values(B, H, W), synth.input_ids(B, ph, cfg.clip.vocab_size), torch.full((B,), ph))
    eng.set_noise(synth.gaussian((B, 4, H // 8, W // 8), 3), synth.gaussian((B, 4, H // 8, W // 8), 4), synth.timesteps(B))
    p0 = eng.params.clone()
    scale = 2.0 ** 40
    skipped = 0
    for _ in range(40):
        eng.step_eager()
        torch.cuda.synchronize()
        if eng.opt_step.item() == 0:
            skipped += 1
            scale *= 0.5
            assert torch.equal(eng.params, p0), "a skipped step must not touch the parameters"
            assert float(eng.scaler[0]) == scale and float(eng.scaler[2]) == 0.0
            assert float(eng.exp_avg.abs().max()) == 0.0
        else:
            break
    print(f"[overflow] {skipped} skipped steps, scale 2^40 -> {float(eng.scaler[0]):.3e}, then opt_step {eng.opt_step.item()}")
    assert skipped >= 1, "a 2^40 loss scale must overflow the f16 gradients"
    assert eng.opt_step.item() == 1 and torch.isfinite(eng.params).all() and not torch.equal(eng.params, p0)


def test_graph_replay_equals_eager_and_trains():
    """hipGraph replay == eager launch list, device RNG advances, loss stays finite, params move."""
    from view_neti_amd import synth
    B, H, W = 2, 64, 64
    cfg, eng, _, _, _, _ = build("tiny", B, H, W, device_rng=True, lr=1e-3, seed=5)
    ph = cfg.clip.vocab_size - 3
    eng.set_batch(synth.pixel_values(B, H, W), synth.input_ids(B, ph, cfg.clip.vocab_size), torch.full((B,), ph))
    eng.capture()
    p0 = eng.params.clone()
    losses, ts = [], []
    for _ in range(6):
        eng.step()
        losses.append(eng.loss())
        ts.append(eng.timesteps.cpu().tolist())
    print("[graph] losses", [f"{l:.4f}" for l in losses], "timesteps", ts[:3])
    assert all(math.isfinite(l) for l in losses)
    assert ts[0] != ts[1], "device RNG must advance between replays"
    assert not torch.equal(p0, eng.params) and torch.isfinite(eng.params).all()
    assert eng.opt_step.item() == 6  # capture() must not leave a warm-up update behind


@pytest.mark.parametrize("cfg_name,B,H,W", [("tiny", 2, 64, 64), ("sd15", 1, 512, 512)], ids=["tiny", "sd15-512"])
def test_two_runs_are_bit_identical(cfg_name, B, H, W):
    """Every reduction of the step is order-independent (GroupNorm statistics: fixed-point integer atomics,
    csrc/common.h vn_fx_*; everything else: fixed-order trees), so the same inputs give the same bits — gradients, the
    prediction, and the parameters after optimizer steps — from one run to the next."""
    from view_neti_amd import synth
    cfg, eng, _, _, _, _ = build(cfg_name, B, H, W, device_rng=False, lr=1e-3)
    ph = cfg.clip.vocab_size - 3
    eng.set_batch(synth.pixel_values(B, H, W), synth.input_ids(B, ph, cfg.clip.vocab_size), torch.full((B,), ph))
    eng.set_noise(synth.gaussian((B, 4, H // 8, W // 8), 3), synth.gaussian((B, 4, H // 8, W // 8), 4), synth.timesteps(B))
    runs = []
    for _ in range(3):
        eng.forward_backward()
        torch.cuda.synchronize()
        runs.append((eng.grads.clone(), eng.unet.pred.clone(), eng.latents.clone(), eng.unet.dctx_k.clone()))
    for r in runs[1:]:
        for a, b, what in zip(runs[0], r, ("mapper gradients", "prediction", "latents", "context gradients")):
            assert torch.equal(a, b), f"{what} differ between two identical runs"
    if cfg_name == "tiny":  # and across engines / graph replays: two fresh engines, four optimizer steps each
        outs = []
        for _ in range(2):
            _, e2, _, _, _, _ = build(cfg_name, B, H, W, device_rng=True, lr=1e-3, seed=9)
            e2.set_batch(synth.pixel_values(B, H, W), synth.input_ids(B, ph, cfg.clip.vocab_size), torch.full((B,), ph))
            e2.capture()
            for _ in range(4):
                e2.step()
            torch.cuda.synchronize()
            outs.append((e2.params.clone(), e2.exp_avg_sq.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_multi_object_mappers_and_segmented_adamw():
    """learnable_mode 3 (BASELINE config 4): several object mappers + one view mapper in one bucket; the
    batch's scene picks the object mapper on the device (graph replay safe); AdamW keeps torch's
    per-parameter semantics (untouched mappers are skipped, touched ones keep decaying with zero grad)."""
    from view_neti_amd import synth
    from view_neti_amd.engine.step import TrainStepEngine
    from view_neti_amd.engine.text import flatten_mapper_state
    from view_neti_amd import sd_config as sc
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    B, H, W, K, lr = 2, 64, 64, 3, 2e-3
    cfg = sc.tiny()
    D = cfg.clip.hidden_size
    gen = torch.Generator().manual_seed(7)
    mk = lambda: {k: v + 0.05 * torch.randn(v.shape, generator=gen) for k, v in init_mapper_state(64, 64, D).items()}
    objs = [mk() for _ in range(K)]
    sdv = mk()
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0)
    w_enc_v = fourier_frequencies([0.03, 2.0] + [0.5] * 12, 64, 0)
    eng = TrainStepEngine(cfg, synth.unet_weights(cfg.unet), synth.vae_weights(cfg.vae), synth.clip_weights(cfg.clip), B,
                          H, W, objs, w_enc, 0.4, 0.2, mapper_view=sdv, w_enc_view=w_enc_v, norm_scale_view=0.35,
                          alpha_view=0.3, lr=lr, seed=11, device_rng=True)
    n = eng.n_obj
    assert eng.n_objects == K and eng.params.numel() == (K + 1) * n
    ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv)
    vparams = synth.gaussian((B, 12), 9).clamp(-1, 1)
    px = synth.pixel_values(B, H, W)
    # torch reference optimizer over [obj_0, obj_1, obj_2, view] with zero_grad(set_to_none=False)
    ref = [torch.nn.Parameter(eng.params[i * n:(i + 1) * n].cpu().clone()) for i in range(K + 1)]
    opt = torch.optim.AdamW(ref, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    eng.capture()
    order = [0, 2, 2, 0, 1]
    for it, k in enumerate(order):
        eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv), vparams, object_index=k)
        before = eng.params.clone()
        eng.step()
        torch.cuda.synchronize()
        g = (eng.grads / (eng.scaler[0] * eng.hyper[5])).cpu()
        assert torch.isfinite(g).all() and g[k * n:(k + 1) * n].abs().sum() > 0 and g[K * n:].abs().sum() > 0
        opt.zero_grad(set_to_none=False)
        ref[k].grad = g[k * n:(k + 1) * n].clone()
        ref[K].grad = g[K * n:].clone()
        opt.step()
        touched = sorted(set(order[: it + 1]))
        for j in range(K):
            moved = not torch.equal(before[j * n:(j + 1) * n], eng.params[j * n:(j + 1) * n])
            assert moved == (j in touched), f"step {it}: object mapper {j} moved={moved}, touched so far {touched}"
    got = eng.params.cpu()
    want = torch.cat([p.detach() for p in ref])
    err = ((got - want).abs().max() / lr).item()
    print(f"[mode3] {K} object mappers + view, order {order}: max |dp|/lr vs torch.optim.AdamW {err:.3e}; "
          f"seg_step {eng.seg_step.tolist()} opt_step {eng.opt_step.item()}")
    assert err < 0.05
    assert eng.seg_step.tolist() == [5, 1, 4] and eng.opt_step.item() == len(order)


# BASELINE config 2 (mode 0, SD-1.5 shapes, 512^2, bs 4) and config 3 (mode 2: object + view mappers, SD-2.1 shapes —
# d=64 heads, 23 CLIP layers, linear projections, v-prediction — on the 384x512 DTU frame)
@pytest.mark.timeout(1500)
@pytest.mark.parametrize("cfg_name,B,H,W,with_view", [("sd15", 4, 512, 512, False), ("sd21", 2, 384, 512, True)],
                         ids=["config2-sd15-512", "config3-sd21-384x512-view"])
def test_full_size_directional_derivative(cfg_name, B, H, W, with_view):
    """A size-independent property beside the oracle comparison below (test_full_size_matches_oracle): the
    mapper gradient produced by the hand-built backward must predict the change of the forward loss along its own
    direction: (L(p+e*d) - L(p-e*d)) / (2e) == g.d,  d = g/|g|.  A 10 % gradient-scale error fails."""
    from view_neti_amd import synth
    cfg, eng, _, _, _, _ = build(cfg_name, B, H, W, with_view, device_rng=False, lr=1e-3)
    ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv if with_view else None)
    vparams = synth.gaussian((B, 12), 9).clamp(-1, 1) if with_view else None
    eng.set_batch(synth.pixel_values(B, H, W), ids, torch.full((B,), ph), torch.full((B,), phv) if with_view else None,
                  vparams)
    eng.set_noise(synth.gaussian((B, 4, H // 8, W // 8), 3), synth.gaussian((B, 4, H // 8, W // 8), 4), synth.timesteps(B))
    eng.forward_backward()
    torch.cuda.synchronize()
    loss0 = eng.loss()
    g = (eng.grads / eng.scaler[0]).clone()
    gn = float(g.norm())
    assert math.isfinite(loss0) and math.isfinite(gn) and gn > 0
    if with_view:
        n = eng.n_all_obj
        assert float(g[:n].norm()) > 0 and float(g[n:].norm()) > 0, "both mappers must receive a gradient"
    d = g / gn
    p0 = eng.params.clone()
    eps = 0.02 / gn if gn > 0.02 else 1.0  # aim at a loss change of ~4e-2 (fp16 noise of the loss is ~1e-4)
    eps = min(eps, 0.05 * float(p0.norm()))  # but stay in the locally linear regime

    def central(e):
        ls = []
        for sgn in (+1.0, -1.0):
            eng.params.copy_(p0 + sgn * e * d)
            eng.forward_backward()
            torch.cuda.synchronize()
            ls.append(eng.loss())
        return (ls[0] - ls[1]) / (2 * e), ls

    # central differences at eps and eps/2, Richardson-extrapolated: D(e) = f' + c e^2 + O(e^4), so (4 D(e/2) - D(e)) / 3
    # removes the third-derivative term (at these step sizes L+ - L0 and L0 - L- differ by tens of percent)
    d1, l1 = central(eps)
    d2, l2 = central(eps / 2)
    eng.params.copy_(p0)
    measured = (4 * d2 - d1) / 3
    ratio = measured / gn
    print(f"[full size {cfg_name} {H}x{W} bs{B} view={with_view}] loss {loss0:.5f} |g| {gn:.4e} eps {eps:.3e}: "
          f"L+ {l1[0]:.5f} L- {l1[1]:.5f}; D(eps) {d1:.4e} D(eps/2) {d2:.4e} -> extrapolated {measured:.4e} vs |g| -> "
          f"ratio {ratio:.3f} (raw {d1 / gn:.3f}, {d2 / gn:.3f})")
    assert l1[0] > loss0 > l1[1], "the loss must rise along +g and fall along -g"
    assert 0.9 < ratio < 1.1


# The north star's gate at the sizes it is quoted on: BASELINE config 2 (SD-1.5 shapes, 512^2) and config 3 (SD-2.1
# shapes, object + view mapper: on the reference's 384x512 DTU frame AND at the 512x512 BASELINE.json states — a 64x64
# latent with d = 64 heads, N = 4096 self-attention and linear projections) against the CPU oracle — the oracle runs a full fp32 forward+backward of these
# shapes in seconds per sample on the host cores (bench.py's cpu_baseline leg times exactly that).
@pytest.mark.timeout(2400)
@pytest.mark.parametrize("cfg_name,B,H,W,with_view", [("sd15", 1, 512, 512, False), ("sd21", 1, 384, 512, True),
                                                      ("sd21", 1, 512, 512, True), ("sd15", 4, 512, 512, False)],
                         ids=["config2-sd15-512-bs1", "config3-sd21-384x512-bs1-view", "config3-sd21-512x512-bs1-view",
                              "config2-sd15-512-bs4"])
def test_full_size_matches_oracle(cfg_name, B, H, W, with_view):
    """eng.forward_backward() vs oracle.sd_ref.train_step_loss(...).backward() on the same fp16-rounded weights, the same
    pixels, noise and timesteps (training/coach.py:165-214).  Bars: loss (predicted-noise MSE) within 1e-3 relative,
    mapper-gradient cosine >= 0.999; the prediction, the latents and every per-layer context are compared too so a
    forward error the backward shares (eps, scale, a mis-packed weight at C=1920/2560, K=23040 split-K, N=4096 d=40
    attention) can not hide."""
    import os
    from oracle import sd_ref as R
    from view_neti_amd import synth
    from view_neti_amd.engine.text import flatten_mapper_state
    if B > 1:
        import psutil
        if psutil.virtual_memory().available < 96 * 2 ** 30:
            pytest.skip("bs=4 fp32 autograd graph of the oracle needs ~60 GiB of host memory")
    torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
    cfg, eng, (uw, vw, cw), sd, w_enc, extra = build(cfg_name, B, H, W, with_view, device_rng=False, lr=1e-3)
    ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv if with_view else None)
    px = synth.pixel_values(B, H, W)
    t = synth.timesteps(B)
    eps = synth.gaussian((B, 4, H // 8, W // 8), 3)
    noise = synth.gaussian((B, 4, H // 8, W // 8), 4)
    vparams = synth.gaussian((B, 12), 9).clamp(-1, 1) if with_view else None
    eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv) if with_view else None, vparams)
    eng.set_noise(eps, noise, t)
    eng.forward_backward()
    torch.cuda.synchronize()
    loss_gpu = eng.loss()
    grads_gpu = (eng.grads / eng.scaler[0]).float().cpu()
    pred_gpu = eng.unet.pred.float().cpu().view(B, H // 8, W // 8, 4).permute(0, 3, 1, 2)  # pixel-major -> NCHW
    lat_gpu = eng.latents.cpu()
    # ---- oracle on the host: fp16-rounded weights (what the GPU holds), fp32 arithmetic ----
    r16 = lambda d: {k: ((v.half().float() if v.dim() >= 2 and "embedding" not in k else v.float()).cpu())
                     for k, v in d.items()}
    uw, vw, cw = r16(uw), r16(vw), r16(cw)
    p_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    view = None
    if with_view:
        p_v = {k: v.clone().requires_grad_(True) for k, v in extra["mapper_view"].items()}
        view = dict(p=p_v, w_enc=extra["w_enc_view"], norm_scale=0.35, placeholder=torch.full((B,), phv),
                    params=vparams, alpha=0.3, unconstrained=False)
    import time
    t0 = time.time()
    loss, aux = R.train_step_loss(cfg, uw, vw, cw, p_o, w_enc, 0.4, px, ids, torch.full((B,), ph), t, eps, noise,
                                  alpha=0.2, view=view)
    loss.backward()
    dt = time.time() - t0
    ref = [flatten_mapper_state({k: v.grad for k, v in p_o.items()})]
    if with_view:
        ref.append(flatten_mapper_state({k: v.grad for k, v in p_v.items()}))
    ref_g = torch.cat(ref)
    rel = abs(loss_gpu - loss.item()) / loss.item()
    cos = torch.nn.functional.cosine_similarity(grads_gpu, ref_g, dim=0).item()
    gerr = ((grads_gpu - ref_g).norm() / ref_g.norm()).item()
    lat = ((lat_gpu - aux["latents"]).norm() / aux["latents"].norm()).item()
    pred_o = aux["pred"].detach()
    pred_rel = ((pred_gpu - pred_o).norm() / pred_o.norm()).item()
    pred_max = (pred_gpu - pred_o).abs().max().item()
    print(f"[full size oracle {cfg_name} {H}x{W} bs{B} view={with_view}] oracle fwd+bwd {dt:.1f}s on "
          f"{torch.get_num_threads()} threads; loss gpu {loss_gpu:.6f} oracle {loss.item():.6f} rel {rel:.2e}; latents "
          f"rel {lat:.2e}; pred rel {pred_rel:.2e} max-abs {pred_max:.2e} (rms {pred_o.pow(2).mean().sqrt():.3f}); "
          f"grad cos {cos:.6f} rel {gerr:.2e} |g| {ref_g.norm():.3e}")
    assert rel < 1e-3, "predicted-noise MSE must match the oracle to 1e-3 relative"
    assert lat < 2e-3 and pred_rel < 1e-2
    assert cos > 0.999 and gerr < 5e-2


@pytest.mark.timeout(2400)
def test_config4_at_size_88_scene_mappers():
    """BASELINE config 4 at its real size on one GPU (train_m3_88scenes.yaml; training/coach.py:505-552, dataset.py:584-600):
    learnable_mode 3, SD-2.1 shapes, 384x512 DTU frames, train_batch_size 3, gradient accumulation 3, 88 object mappers
    (hidden 64, 1024-wide CLIP: 141 696 floats each) + the view mapper in ONE bucket (12.6 M floats, 89 AdamW segments).
    Six optimisation steps (18 captured micro-steps) over a seeded scene sequence; the bucket must follow
    torch.optim.AdamW fed with the engine's own gradients (untouched scenes never move, touched ones keep decaying)."""
    import time
    import numpy as np
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.step import TrainStepEngine
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    B, H, W, K, accum, lr = 3, 384, 512, 88, 3, 1e-3 * 3 * 3
    cfg = sc.CONFIGS["sd21"]()
    D = cfg.clip.hidden_size
    assert D == 1024
    dev = "cuda"
    uw, vw, cw = synth.unet_weights(cfg.unet, device=dev), synth.vae_weights(cfg.vae, device=dev), synth.clip_weights(cfg.clip, device=dev)
    gen = torch.Generator().manual_seed(7)
    base = init_mapper_state(64, 64, D)
    mk = lambda: {k: v + 0.02 * torch.randn(v.shape, generator=gen) for k, v in base.items()}
    objs = [mk() for _ in range(K)]
    sdv = mk()
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0)
    w_enc_v = fourier_frequencies([0.03, 2.0] + [2.0] * 12, 64, 0)
    eng = TrainStepEngine(cfg, uw, vw, cw, B, H, W, objs, w_enc, 0.4, 5.0, mapper_view=sdv, w_enc_view=w_enc_v,
                          norm_scale_view=0.35, alpha_view=5.0, lr=lr, seed=11, device_rng=True, grad_accum=accum)
    del uw, vw, cw
    n = eng.n_obj
    assert n == 141696 and eng.n_objects == K and eng.params.numel() == (K + 1) * n == 12_610_944
    print(f"[config 4] bucket {(K + 1) * n * 4 / 2 ** 20:.1f} MiB ({K} object mappers + view), engine {eng.memory_bytes() / 2 ** 30:.1f} GiB")
    ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv)
    vparams = synth.gaussian((B, 12), 9).clamp(-1, 1)
    px = synth.pixel_values(B, H, W)
    ref = [torch.nn.Parameter(eng.params[i * n:(i + 1) * n].cpu().clone()) for i in range(K + 1)]
    opt = torch.optim.AdamW(ref, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    np.random.seed(0)  # dataset.py:584-600: np.random.choice(len(train_data_subsets)) once per optimisation step
    scenes = [int(np.random.choice(K)) for _ in range(6)]
    scenes[3] = scenes[0]  # revisit a scene: its Adam moments must have kept decaying in between
    eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv), vparams, object_index=scenes[0])
    eng.capture()
    t0 = None
    for it, k in enumerate(scenes):
        if it == 1:
            torch.cuda.synchronize()
            t0 = time.time()
        before = eng.params.clone()
        for _ in range(accum):
            eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv), vparams, object_index=k)
            stepped = eng.step()
        assert stepped
        g = (eng.grads / (eng.scaler[0] * eng.hyper[5])).cpu()
        assert torch.isfinite(g).all() and g[k * n:(k + 1) * n].abs().sum() > 0 and g[K * n:].abs().sum() > 0
        # (segments of scenes trained earlier keep their last gradient in the bucket; the optimizer reads the active
        #  scene's segment and the view mapper only — the moved / not-moved check below is the observable)
        opt.zero_grad(set_to_none=False)
        ref[k].grad = g[k * n:(k + 1) * n].clone()
        ref[K].grad = g[K * n:].clone()
        opt.step()
        touched = set(scenes[: it + 1])
        moved = [(not torch.equal(before[j * n:(j + 1) * n], eng.params[j * n:(j + 1) * n])) for j in range(K)]
        assert [j for j in range(K) if moved[j]] == sorted(touched), (it, sorted(touched))
    torch.cuda.synchronize()
    ms = (time.time() - t0) / (5 * accum) * 1e3
    got = eng.params.cpu()
    want = torch.cat([p.detach() for p in ref])
    err = ((got - want).abs().max() / lr).item()
    print(f"[config 4] scenes {scenes}: max |dp|/lr vs torch.optim.AdamW {err:.3e}; {ms:.1f} ms per micro-step "
          f"(bs {B}, 384x512, sd21) = {1e3 / ms:.1f} micro-steps/s; opt_step {eng.opt_step.item()}")
    assert err < 0.05 and eng.opt_step.item() == len(scenes)


def test_vae_moment_cache_is_bit_identical_and_skips_the_encoder():
    """data.cache_vae_moments (an extension, never the benchmark's configuration): with a deterministic
    dataset the posterior moments of an image (vae.encode(...).latent_dist, training/coach.py:165) are kept in HBM and only
    `.sample()` is re-drawn.  Same images, same noise => the same step bit for bit, with and without the cache, eager and
    captured; a batch holding an image the cache has not seen runs the encoder."""
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.step import TrainStepEngine
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    cfg = sc.tiny()
    B, H, W = 2, 64, 64
    torch.manual_seed(0)
    sd = init_mapper_state(64, 64, cfg.clip.hidden_size)
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0)
    images = [synth.gaussian((3, H, W), 40 + i).clamp(-1, 1) for i in range(5)]
    ph = cfg.clip.vocab_size - 3
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size)
    order = [(0, 1), (2, 3), (1, 0), (4, 2), (3, 3), (0, 4)]

    def run(n_cache, graph):
        eng = TrainStepEngine(cfg, synth.unet_weights(cfg.unet), synth.vae_weights(cfg.vae), synth.clip_weights(cfg.clip), B,
                              H, W, sd, w_enc, 0.4, 0.2, lr=3e-3, seed=11, moment_cache_images=n_cache)
        vae_runs = [0]
        fwd = eng.vae.forward
        eng.vae.forward = lambda: (vae_runs.__setitem__(0, vae_runs[0] + 1), fwd())[1]
        losses = []
        for step, pair in enumerate(order):
            px = torch.stack([images[i] for i in pair])
            eng.set_batch(px, ids, torch.full((B,), ph), image_idx=torch.tensor(pair) if n_cache else None)
            if graph and step == 0:
                eng.capture()
            eng.step()
            losses.append(eng.loss())
        torch.cuda.synchronize()
        return eng.params.clone(), losses, vae_runs[0], eng

    p0, l0, runs0, _ = run(0, False)
    p1, l1, runs1, e1 = run(5, False)
    assert runs0 == len(order)
    # batches 0, 1 and 3 bring new images; (1, 0), (3, 3) and (0, 4) are served from the cache
    assert runs1 == 3 and sorted(e1._cached_images) == [0, 1, 2, 3, 4]
    assert l0 == l1 and torch.equal(p0, p1), "the cached moments must reproduce the uncached step bit for bit"
    p2, l2, _, e2 = run(5, True)
    assert e2.graph_a_c is not None
    assert l2 == l0 and torch.equal(p2, p0), "captured: full-step graph and cached-step graph interleaved"
    with pytest.raises(ValueError, match="image_idx"):
        e2.set_batch(torch.stack(images[:2]), ids, torch.full((B,), ph))
    with pytest.raises(ValueError, match="outside"):
        e2.set_batch(torch.stack(images[:2]), ids, torch.full((B,), ph), image_idx=[0, 5])
