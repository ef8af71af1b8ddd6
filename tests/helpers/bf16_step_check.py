"""Run in its OWN process with VNETI_PRECISION=bf16 (a process computes in one 16-bit format, view_neti_amd/lib.py):
the whole train step of libvneti_hip_bf16.so — the reference's `mixed_precision: bf16` branch, training/coach.py:796-802 —
against the CPU oracle on bf16-rounded weights (fp32 arithmetic), plus one AdamW step.  Prints one JSON line.
    python tests/helpers/bf16_step_check.py tiny|tiny21|sd15 [B H W]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from view_neti_amd import lib  # noqa: E402

assert lib.precision() == "bf16", "start this script with VNETI_PRECISION=bf16"
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 64, 64)
from oracle import sd_ref as R  # noqa: E402
from test_step_gpu import build  # noqa: E402
from view_neti_amd import synth  # noqa: E402
from view_neti_amd.engine.text import flatten_mapper_state  # noqa: E402

with_view = name != "sd15"
torch.set_num_threads(min(32, len(os.sched_getaffinity(0))))
cfg, eng, (uw, vw, cw), sd, w_enc, extra = build(name, B, H, W, with_view, device_rng=False, lr=4e-3)
assert eng.unet.pred.dtype == torch.bfloat16 and float(eng.scaler[0]) == 1.0  # bf16 buffers, no loss scaling (accelerate)
ph, phv = cfg.clip.vocab_size - 3, cfg.clip.vocab_size - 4
ids = synth.input_ids(B, ph, cfg.clip.vocab_size, view_placeholder_id=phv if with_view else None)
px, t = synth.pixel_values(B, H, W), synth.timesteps(B)
eps, noise = synth.gaussian((B, 4, H // 8, W // 8), 3), synth.gaussian((B, 4, H // 8, W // 8), 4)
vparams = synth.gaussian((B, 12), 9).clamp(-1, 1) if with_view else None
eng.set_batch(px, ids, torch.full((B,), ph), torch.full((B,), phv) if with_view else None, vparams)
eng.set_noise(eps, noise, t)
p0 = eng.params.clone()
eng.forward_backward()
torch.cuda.synchronize()
loss_gpu = eng.loss()
grads_gpu = (eng.grads / eng.scaler[0]).float().cpu()
rbf = lambda d: {k: (v.cpu().bfloat16().float() if v.dim() >= 2 and "embedding" not in k else v.cpu().float()) for k, v in d.items()}
p_o = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
view = None
if with_view:
    p_v = {k: v.clone().requires_grad_(True) for k, v in extra["mapper_view"].items()}
    view = dict(p=p_v, w_enc=extra["w_enc_view"], norm_scale=0.35, placeholder=torch.full((B,), phv), params=vparams, alpha=0.3)
loss, aux = R.train_step_loss(cfg, rbf(uw), rbf(vw), rbf(cw), p_o, w_enc, 0.4, px, ids, torch.full((B,), ph), t, eps, noise,
                              alpha=0.2, view=view)
loss.backward()
ref = [flatten_mapper_state({k: v.grad for k, v in p_o.items()})]
if with_view:
    ref.append(flatten_mapper_state({k: v.grad for k, v in p_v.items()}))
ref_g = torch.cat(ref)
eng.optimizer_step()
torch.cuda.synchronize()
p1, _, _ = R.adamw_step(p0.cpu(), grads_gpu, torch.zeros_like(ref_g), torch.zeros_like(ref_g), 1, 4e-3)
print(json.dumps({
    "config": name, "B": B, "H": H, "W": W, "loss_gpu": loss_gpu, "loss_oracle": loss.item(),
    "loss_rel": abs(loss_gpu - loss.item()) / loss.item(),
    "latents_rel": ((eng.latents.cpu() - aux["latents"]).norm() / aux["latents"].norm()).item(),
    "grad_cos": torch.nn.functional.cosine_similarity(grads_gpu, ref_g, dim=0).item(),
    "grad_rel": ((grads_gpu - ref_g).norm() / ref_g.norm()).item(),
    "adamw_dev_over_lr": ((eng.params.cpu() - p1).abs().max() / 4e-3).item(), "opt_step": int(eng.opt_step.item()),
    "finite": bool(torch.isfinite(eng.params).all())}))
