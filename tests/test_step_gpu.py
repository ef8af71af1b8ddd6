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
    gpu = cfg_name != "tiny"
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


@pytest.mark.parametrize("with_view", [False, True])
def test_train_step_tiny_matches_oracle(with_view):
    from oracle import sd_ref as R
    from view_neti_amd import synth
    from view_neti_amd.engine.text import flatten_mapper_state
    B, H, W = 2, 64, 64
    lr = 4e-3
    cfg, eng, (uw, vw, cw), sd, w_enc, extra = build("tiny", B, H, W, with_view, device_rng=False, lr=lr)
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
                    params=vparams, alpha=0.3)
    loss, aux = R.train_step_loss(cfg, r16(uw), r16(vw), r16(cw), p_o, w_enc, 0.4, px, ids, torch.full((B,), ph), t,
                                  eps, noise, alpha=0.2, view=view)
    loss.backward()
    ref = [flatten_mapper_state({k: v.grad for k, v in p_o.items()})]
    if with_view:
        ref.append(flatten_mapper_state({k: v.grad for k, v in p_v.items()}))
    ref_g = torch.cat(ref)
    rel = abs(loss_gpu - loss.item()) / loss.item()
    cos = torch.nn.functional.cosine_similarity(grads_gpu, ref_g, dim=0).item()
    gerr = ((grads_gpu - ref_g).norm() / ref_g.norm()).item()
    lat = ((eng.latents.cpu() - aux["latents"]).norm() / aux["latents"].norm()).item()
    print(f"[step tiny view={with_view}] loss gpu {loss_gpu:.6f} oracle {loss.item():.6f} rel {rel:.2e}; "
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
