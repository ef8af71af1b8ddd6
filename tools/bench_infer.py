"""Dev tool: inference throughput (BASELINE.json config 5: learnable_mode 5 style single-view generation,
SD-2.1 shapes, 768x768 fp16, 50 sampler steps).  Prints one JSON line (images/s and s/image)."""
import os
os.environ.setdefault("VNETI_ALLOW_SYNTHETIC_WEIGHTS", "1")  # dev tool: synthetic SD-shaped weights on purpose
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from view_neti_amd import sd_config as sc, synth
from view_neti_amd.engine.infer import InferenceEngine
from view_neti_amd.mapper import fourier_frequencies, init_mapper_state

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="sd21"); ap.add_argument("--resolution", type=int, default=768)
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--sampler", default="ddim"); ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--height", type=int, default=0); ap.add_argument("--width", type=int, default=0)  # default: resolution^2
a = ap.parse_args()
a.height, a.width = a.height or a.resolution, a.width or a.resolution
cfg = sc.CONFIGS[a.model]()
dev = "cuda"
uw, dw, cw = synth.unet_weights(cfg.unet, device=dev), synth.vae_decoder_weights(cfg.vae, device=dev), synth.clip_weights(cfg.clip, device=dev)
D = cfg.clip.hidden_size
torch.manual_seed(0)
w_enc = fourier_frequencies([0.03, 2.0], 64, 0); sdo = init_mapper_state(64, 64, D)
w_enc_v = fourier_frequencies([0.03, 2.0] + [0.5] * 12, 64, 0); sdv = init_mapper_state(64, 64, D)
norm = float(cw["text_model.embeddings.token_embedding.weight"][:1000].float().norm(dim=1).mean())
t0 = time.time()
eng = InferenceEngine(cfg, uw, dw, cw, a.batch, a.height, a.width, sdo, w_enc, norm, 5.0, mapper_view=sdv,
                      w_enc_view=w_enc_v, norm_scale_view=norm, alpha_view=5.0)
del uw, dw, cw
ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
ids = synth.input_ids(a.batch, ph, cfg.clip.vocab_size, view_placeholder_id=phv)
neg = synth.input_ids(1, ph, cfg.clip.vocab_size); neg[neg == ph] = 7
eng.set_negative_prompt(neg)
eng.set_prompt(ids, torch.full((a.batch,), ph), torch.full((a.batch,), phv), synth.gaussian((a.batch, 12), 9).clamp(-1, 1))
lat = synth.gaussian((a.batch, 4, a.height // 8, a.width // 8), 17)
build_s = time.time() - t0
img = eng.generate(lat, a.steps, 7.5, a.sampler); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    img = eng.generate(lat, a.steps, 7.5, a.sampler)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
print(json.dumps({"metric": f"NeTI inference images/s ({a.model} {a.width}x{a.height}, {a.sampler}-{a.steps}, CFG, bs={a.batch})",
                  "value": a.batch / dt, "unit": "images/s", "s_per_image": dt / a.batch, "ms_per_sampler_step": dt / a.steps * 1e3,
                  "engine_gib": eng.memory_bytes() / 2 ** 30, "build_s": build_s, "image_finite": bool(torch.isfinite(img).all()),
                  "image_mean": float(img.mean())}))
