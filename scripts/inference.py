"""Generate images with the mappers of a finished run (cf. the reference's scripts/inference.py /
training/inference_dtu.py; minimal: prompts x seeds -> PNG files).

    python scripts/inference.py --exp_dir results/train --prompt "<view_dtu12d_cam22_…>. A photo of a <object>" \
        --seeds 0 1 --steps 30 --guidance 7.5 --out out/
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from view_neti_amd.compat.inference import load_inference  # noqa: E402
from view_neti_amd.compat.sd_pipeline_call import sd_pipeline_call  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp_dir", required=True)
    ap.add_argument("--mapper", default="mapper-final")
    ap.add_argument("--prompt", action="append", required=True)
    ap.add_argument("--seeds", type=int, nargs="+", default=[0])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--guidance", type=float, default=7.5)
    ap.add_argument("--sampler", default="dpm++2m", choices=["dpm++2m", "ddim"])
    ap.add_argument("--truncation_idx", type=int, default=None)
    ap.add_argument("--out", default="inference_out")
    a = ap.parse_args()
    pipe, pm = load_inference(a.exp_dir, a.mapper, batch=1, sampler=a.sampler)
    os.makedirs(a.out, exist_ok=True)
    for pi, prompt in enumerate(a.prompt):
        emb = pm.embed_prompt(prompt, truncation_idx=a.truncation_idx)
        for seed in a.seeds:
            out = sd_pipeline_call(pipe, emb, num_inference_steps=a.steps, guidance_scale=a.guidance,
                                   generator=torch.Generator().manual_seed(seed))
            path = os.path.join(a.out, f"p{pi}_s{seed}.png")
            out.images[0].save(path)
            print(path)


if __name__ == "__main__":
    main()
