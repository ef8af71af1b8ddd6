"""Test infrastructure (like everything under oracle/): deterministic stand-ins for the two diffusers modules the
reference's inference loop calls — a UNet and a scheduler — so that the REAL `/root/reference/sd_pipeline_call.py:73-98`
loop (imported by oracle/make_golden.py, fixture G10) and the restatement `oracle/sd_ref.sd_pipeline_call` can be driven
with the very same pair and compared call by call.  Nothing here is shipped or measured."""
from __future__ import annotations

import torch


class ToyUNet:
    """`unet(latents, t, encoder_hidden_states=..., cross_attention_kwargs=...).sample`, linear in the latents and
    sensitive to WHICH conditioning it was handed: a tensor (the negative prompt) or an XTI dict, of which it consumes the
    16 per-layer (context, bypass) pairs in order through `this_idx`, exactly as the 16 XTIAttenProc instances of one
    forward do (models/xti_attention_processor.py:27-41).  Every call is logged."""

    in_channels = 4

    class config:
        sample_size = 8

    def __init__(self, n_layers: int = 16):
        self.n_layers = n_layers
        self.calls = []  # (t, kind, tag): kind 0 = tensor conditioning, 1 = dict; tag = the dict's "_tag" entry or -1
        g = torch.Generator().manual_seed(1234)
        self.wk = torch.rand(n_layers, generator=g) - 0.5
        self.wv = torch.rand(n_layers, generator=g) - 0.5

    def summarize(self, ehs):
        if isinstance(ehs, dict):
            acc = 0.0
            for _ in range(self.n_layers):
                i = ehs["this_idx"]
                k, v = ehs[f"CONTEXT_TENSOR_{i}"], ehs[f"CONTEXT_TENSOR_BYPASS_{i}"]
                acc = acc + self.wk[i] * k.double().mean() + self.wv[i] * v.double().pow(2).mean()
                ehs["this_idx"] = (i + 1) % self.n_layers
            return acc
        return ehs.double().mean() * 0.37

    def __call__(self, latents, t, encoder_hidden_states=None, cross_attention_kwargs=None):
        ehs = encoder_hidden_states
        kind = 1 if isinstance(ehs, dict) else 0
        tag = int(ehs.get("_tag", -1)) if kind else -1
        self.calls.append((int(t), kind, tag))
        c = self.summarize(ehs)
        out = 0.8 * latents + (0.1 * float(t) / 1000.0) * latents.flip(-1) + float(c)
        return type("UNetOutput", (), {"sample": out})()


def toy_decode(latents):
    """stands where `pipeline.decode_latents` (diffusers) stands: any deterministic function of the final latents"""
    return (latents / 0.18215).tanh().mul(0.5).add(0.5).clamp(0, 1).permute(0, 2, 3, 1)
