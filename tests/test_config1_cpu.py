"""BASELINE.json configs[0]: learnable_mode 0, colorful_teapot, SD-1.5 shapes, 256x256, bs=1, 2 optimisation steps on the
CPU path — plumbing only, no GPU.  Here the CPU path is the oracle (oracle/sd_ref.py; diffusers is not installable): the
reference's dataset class restated in compat/ reads two of the reference's own teapot photographs (fixtures under
tests/golden/colorful_teapot, data files of /root/reference/data/datasets_mode0/colorful_teapot), the oracle runs
VAE -> noise -> 16 x text -> UNet -> MSE -> backward, AdamW moves the mapper (training/coach.py:154-231)."""
import math
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def test_config1_teapot_two_steps_on_the_cpu_oracle():
    from oracle import sd_ref as R
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.compat.dataset import TextualInversionDataset
    from view_neti_amd.compat.tokenizer import HashTokenizer
    from view_neti_amd.engine.text import flatten_mapper_state, unflatten_mapper_state
    from view_neti_amd.mapper import fourier_frequencies, init_mapper_state
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cfg = sc.CONFIGS["sd15"]()
    tk = HashTokenizer()
    tk.add_tokens(["<teapot>"])
    ds = TextualInversionDataset(os.path.join(HERE, "golden", "colorful_teapot"), tk, learnable_mode=0, size=256,
                                 placeholder_object_token="<teapot>")
    assert ds.num_images == 2
    uw, vw, cw = synth.unet_weights(cfg.unet), synth.vae_weights(cfg.vae), synth.clip_weights(cfg.clip)
    w_enc = fourier_frequencies([0.03, 2.0], 64, 0)
    sd = {k: v.clone().requires_grad_(True) for k, v in init_mapper_state(64, 64, cfg.clip.hidden_size).items()}
    names = list(sd)
    lr, B = 1e-3, 1
    m = torch.zeros(sum(v.numel() for v in sd.values()))
    v2 = torch.zeros_like(m)
    losses = []
    for step in range(2):
        ex = ds[step]
        px = ex["pixel_values"][None]
        assert px.shape == (1, 3, 256, 256)
        ids = ex["input_ids"][None] % cfg.clip.vocab_size
        ph = ex["input_ids_placeholder_object"].reshape(1) % cfg.clip.vocab_size
        t = synth.timesteps(B, seed=step)
        eps, noise = synth.gaussian((B, 4, 32, 32), 3 + step), synth.gaussian((B, 4, 32, 32), 5 + step)
        for p in sd.values():
            p.grad = None
        loss, aux = R.train_step_loss(cfg, uw, vw, cw, sd, w_enc, 0.4, px, ids, ph, t, eps, noise, alpha=0.2)
        loss.backward()
        losses.append(loss.item())
        g = flatten_mapper_state({k: sd[k].grad for k in names})
        assert torch.isfinite(g).all() and g.norm() > 0
        p0 = flatten_mapper_state({k: sd[k].detach() for k in names})
        p1, m, v2 = R.adamw_step(p0, g, m, v2, step + 1, lr)
        assert not torch.equal(p0, p1)
        new = unflatten_mapper_state(p1, 64, 64, 2 * cfg.clip.hidden_size)
        with torch.no_grad():
            for k in names:
                sd[k].copy_(new[k])
    assert all(math.isfinite(l) and l > 0 for l in losses)
    assert aux["latents"].shape == (1, 4, 32, 32) and aux["pred"].shape == (1, 4, 32, 32)

