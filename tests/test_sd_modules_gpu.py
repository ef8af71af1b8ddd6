"""Seam B (module-call protocol): `vae.encode(x).latent_dist.sample()`, `vae.config.scaling_factor`,
`noise_scheduler.add_noise / get_velocity / .config`, `unet(sample, timesteps, ctx_dict).sample` served by the HIP engines
(compat/sd_modules.py) and driven in the exact order of the reference's loop body (training/coach.py:165-214), against
(a) `TrainStepEngine` on the same inputs — the forward must agree bit for bit, it is the same launch schedule — and
(b) the CPU oracle (loss within 1e-3 relative, context gradients within 1e-2 relative / cosine 0.999)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg_name,with_view", [("tiny", False), ("tiny21", True)])
def test_module_call_protocol_vae_scheduler_unet(cfg_name, with_view):
    from oracle import sd_ref as R
    from test_step_gpu import build
    from view_neti_amd import synth
    from view_neti_amd.compat.sd_modules import HipAutoencoderKL, HipDDPMScheduler, HipUNet2DConditionModel
    B, H, W = 2, 64, 64
    dev = "cuda"
    cfg, eng, (uw, vw, cw), sd, w_enc, extra = build(cfg_name, B, H, W, with_view, device_rng=False, lr=1e-3)
    vae = HipAutoencoderKL(cfg.vae, vw, B, H, W)
    sched = HipDDPMScheduler(cfg.ddpm)
    unet = HipUNet2DConditionModel(cfg.unet, uw, B, H // 8, W // 8, cfg.clip.max_positions)
    assert sched.config.num_train_timesteps == 1000 and vae.config.scaling_factor == cfg.vae.scaling_factor
    ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
    ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv if with_view else None)
    px = synth.pixel_values(B, H, W)
    vparams = synth.gaussian((B, 12), 9).clamp(-1, 1) if with_view else None
    torch.manual_seed(123)

    # ---- the reference's loop body, on the adapters (coach.py:165-214) ----
    latents = vae.encode(px.to(dev)).latent_dist.sample().detach()
    latents = latents * vae.config.scaling_factor
    noise = torch.randn_like(latents)
    bsz = latents.shape[0]
    timesteps = torch.randint(low=0, high=sched.config.num_train_timesteps, size=(bsz,), device=latents.device).long()
    noisy_latents = sched.add_noise(latents, noise, timesteps)
    # text conditioning from the engine's own text pass (the text seam has its own test, tests/test_text_gpu.py); as leaf
    # tensors so that the UNet adapter's backward has somewhere to deliver the context gradients
    eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv) if with_view else None, vparams)
    eng.set_noise(vae.last_eps, noise, timesteps)
    eng.forward_backward()  # the engine's whole step on the same randomness: reference values for everything below
    torch.cuda.synchronize()
    L, D, nl = cfg.clip.max_positions, cfg.clip.hidden_size, cfg.unet.n_cross_layers
    hs = {"this_idx": 0}
    for i in range(nl):
        hs[f"CONTEXT_TENSOR_{i}"] = eng.unet.ctx_k[i].view(B, L, D).clone().requires_grad_(True)
        hs[f"CONTEXT_TENSOR_BYPASS_{i}"] = eng.unet.ctx_v[i].view(B, L, D).clone().requires_grad_(True)
    model_pred = unet(noisy_latents, timesteps, hs).sample
    if sched.config.prediction_type == "epsilon":
        target = noise
    else:
        target = sched.get_velocity(latents, noise, timesteps)
    loss = F.mse_loss(model_pred.float(), target.float(), reduction="mean")
    scale = float(eng.scaler[0])
    (loss * scale).backward()  # GradScaler-scaled backward (accelerator.backward under fp16)
    torch.cuda.synchronize()

    # ---- (a) same launch schedules as TrainStepEngine: forward values bit for bit ----
    assert torch.equal(latents, eng.latents), "latent_dist.sample() * scaling_factor"
    assert torch.equal(noisy_latents, eng.unet.x_in), "add_noise"
    assert torch.equal(target, eng.target), "loss target"
    pred_eng = eng.unet.pred.view(B, H // 8, W // 8, 4).permute(0, 3, 1, 2)
    assert torch.equal(model_pred.detach(), pred_eng), "unet(...).sample"
    rel_l = abs(loss.item() - eng.loss()) / eng.loss()
    gk = torch.stack([hs[f"CONTEXT_TENSOR_{i}"].grad.reshape(B * L, D) for i in range(nl)]).float()
    gv = torch.stack([hs[f"CONTEXT_TENSOR_BYPASS_{i}"].grad.reshape(B * L, D) for i in range(nl)]).float()
    ek, ev = eng.unet.dctx_k.float(), eng.unet.dctx_v.float()
    rk = ((gk - ek).norm() / ek.norm()).item()
    rv = ((gv - ev).norm() / ev.norm()).item()
    print(f"[seam B {cfg_name}] loss {loss.item():.6f} vs engine {eng.loss():.6f} (rel {rel_l:.1e}); context gradients vs engine: "
          f"K rel {rk:.2e}, V rel {rv:.2e}")
    # (the backward seeds differ in the last f32 bit: torch's mse backward vs the fused mse_loss_grad kernel)
    assert rel_l < 1e-5 and rk < 2e-3 and rv < 2e-3

    # ---- (b) the oracle on the same inputs ----
    r16 = lambda d: {k: (v.half().float() if v.dim() >= 2 and "embedding" not in k else v) for k, v in d.items()}
    ctx = {"this_idx": 0}
    leaves = []
    for i in range(nl):
        k = hs[f"CONTEXT_TENSOR_{i}"].detach().float().cpu().requires_grad_(True)
        v = hs[f"CONTEXT_TENSOR_BYPASS_{i}"].detach().float().cpu().requires_grad_(True)
        ctx[f"CONTEXT_TENSOR_{i}"], ctx[f"CONTEXT_TENSOR_BYPASS_{i}"] = k, v
        leaves += [k, v]
    pred_o = R.unet_forward(r16(uw), cfg.unet, noisy_latents.cpu(), timesteps.cpu(), ctx)
    loss_o = F.mse_loss(pred_o, target.cpu())
    loss_o.backward()
    rel = abs(loss.item() - loss_o.item()) / loss_o.item()
    go_k = torch.stack([ctx[f"CONTEXT_TENSOR_{i}"].grad.reshape(B * L, D) for i in range(nl)])
    go_v = torch.stack([ctx[f"CONTEXT_TENSOR_BYPASS_{i}"].grad.reshape(B * L, D) for i in range(nl)])
    g_all = torch.cat([gk.cpu().flatten(), gv.cpu().flatten()]) / scale
    o_all = torch.cat([go_k.flatten(), go_v.flatten()])
    cos = F.cosine_similarity(g_all, o_all, dim=0).item()
    gerr = ((g_all - o_all).norm() / o_all.norm()).item()
    print(f"[seam B {cfg_name}] vs oracle: loss rel {rel:.2e}; context-gradient cosine {cos:.6f} rel {gerr:.2e}")
    assert rel < 1e-3 and cos > 0.999 and gerr < 3e-2
    assert hs["this_idx"] == 0
