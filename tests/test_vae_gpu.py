"""VAE encoder engine vs the CPU oracle (moments = quant_conv(encoder(x)))."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 64, 128)])
def test_vae_encoder_tiny(B, H, W):
    from oracle import sd_ref as R
    from view_neti_amd import sd_config as sc, synth
    from view_neti_amd.engine.vae import VAEEncoderEngine
    cfg = sc.tiny().vae
    w = {k: v.half().float() for k, v in synth.vae_weights(cfg).items()}
    eng = VAEEncoderEngine(cfg, w, B, H, W)
    x = synth.pixel_values(B, H, W)
    eng.x_in.copy_(x)
    eng.forward()
    torch.cuda.synchronize()
    got = eng.moments.float().cpu().view(B, H // 8, W // 8, 8).permute(0, 3, 1, 2)
    ref = R.vae_encode_moments(w, cfg, x.half().float())
    rel = ((got - ref).norm() / ref.norm()).item()
    print(f"[vae tiny {B}x{H}x{W}] moments rel err {rel:.3e} (ref std {ref.std():.3f}), {len(eng.fwd)} launches")
    assert math.isfinite(rel) and rel < 1e-2
