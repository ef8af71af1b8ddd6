"""Deterministic, library-version-independent synthetic weights and inputs.

No SD checkpoints, tokenizer files or datasets exist on the build or GPU boxes (SURVEY.md §7 hard
part 1), so throughput and parity are measured on SD-shaped random tensors.  Values come from an
integer counter hash (not torch.randn), so the build container, the GPU box and any future torch
version regenerate bit-identical fp32 tensors from (seed, tensor name).

Scaling keeps activations O(1) through ~60 layers: Linear/conv weights U(-1,1)*sqrt(3/fan_in)
(unit gain), norm gains 1 + 0.02u, biases / norm shifts 0.02u, embeddings 0.02*sqrt(3)*u.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch

from . import sd_config as sc


def _hash_u32(idx: np.ndarray, seed: int) -> np.ndarray:
    """lowbias32-style avalanche of (index, seed) on uint32 lanes."""
    x = idx.astype(np.uint32) * np.uint32(0x9E3779B1) + np.uint32(seed & 0xFFFFFFFF)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def uniform_pm1(n: int, seed: int) -> np.ndarray:
    """n floats in [-1, 1), float32, exactly reproducible."""
    out = np.empty(n, dtype=np.float32)
    step = 1 << 24
    for s in range(0, n, step):
        e = min(n, s + step)
        with np.errstate(over="ignore"):
            h = _hash_u32(np.arange(s, e, dtype=np.uint64) & 0xFFFFFFFF, seed + (s >> 24) * 0x632BE5AB)
        out[s:e] = (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return out


def normal_like(n: int, seed: int) -> np.ndarray:
    """approx N(0,1) via sum of 4 uniforms (Irwin-Hall), reproducible; used for noise/eps inputs."""
    acc = np.zeros(n, dtype=np.float32)
    for k in range(4):
        acc += uniform_pm1(n, seed * 4 + k + 17)
    return acc * np.float32(np.sqrt(3.0 / 4.0))


def uniform_pm1_torch(n: int, seed: int, device="cpu") -> torch.Tensor:
    """Bit-identical to uniform_pm1 but in torch int64 arithmetic (wraps mod 2^64, low 32 bits
    exact), so it can run multi-threaded on CPU or on the GPU."""
    M = 0xFFFFFFFF
    out = torch.empty(n, dtype=torch.float32, device=device)
    step = 1 << 24
    for s in range(0, n, step):
        e = min(n, s + step)
        sd = (seed + (s >> 24) * 0x632BE5AB) & M
        x = torch.arange(s, e, dtype=torch.int64, device=device)
        x = (x * 0x9E3779B1 + sd) & M
        x = x ^ (x >> 16)
        x = (x * 0x7FEB352D) & M
        x = x ^ (x >> 15)
        x = (x * 0x846CA68B) & M
        x = x ^ (x >> 16)
        out[s:e] = (x >> 8).to(torch.float32) * (2.0 ** -23) - 1.0
    return out


def _name_seed(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def make_tensor(name: str, shape, seed: int, device="cpu") -> torch.Tensor:
    n = int(np.prod(shape))
    u = uniform_pm1_torch(n, _name_seed(name, seed), device)
    if name.endswith("embedding.weight"):
        u = u * float(np.float32(0.02 * np.sqrt(3.0)))
    elif name.endswith(".bias"):
        u = u * float(np.float32(0.02))
    elif len(shape) == 1:  # norm gains
        u = 1.0 + float(np.float32(0.02)) * u
    else:
        fan_in = int(np.prod(shape[1:]))
        u = u * float(np.float32(np.sqrt(3.0 / fan_in)))
    return u.reshape(shape)


def make_state_dict(shapes: sc.Shapes, seed: int, device="cpu") -> Dict[str, torch.Tensor]:
    return {name: make_tensor(name, shape, seed, device) for name, shape in shapes.items()}


def unet_weights(cfg: sc.UNetConfig, seed: int = 1234, device="cpu"):
    return make_state_dict(sc.unet_shapes(cfg), seed, device)


def vae_weights(cfg: sc.VAEConfig, seed: int = 1234, device="cpu"):
    return make_state_dict(sc.vae_encoder_shapes(cfg), seed + 1, device)


def vae_decoder_weights(cfg: sc.VAEConfig, seed: int = 1234, device="cpu"):
    return make_state_dict(sc.vae_decoder_shapes(cfg), seed + 3, device)


def clip_weights(cfg: sc.CLIPTextConfig, seed: int = 1234, device="cpu"):
    return make_state_dict(sc.clip_text_shapes(cfg), seed + 2, device)


# ----------------------------------------------------------------------------------------------
# synthetic step inputs
# ----------------------------------------------------------------------------------------------
def pixel_values(batch: int, height: int, width: int, seed: int = 1) -> torch.Tensor:
    """uint8-uniform image -> [-1, 1] f32, NCHW like TextualInversionDataset.__getitem__
    (training/dataset.py:733-737: image/127.5 - 1, HWC->CHW)."""
    n = batch * 3 * height * width
    h = _hash_u32(np.arange(n, dtype=np.uint64), seed * 7919 + 11) >> np.uint32(24)
    img = h.astype(np.float32) / np.float32(127.5) - np.float32(1.0)
    return torch.from_numpy(img.reshape(batch, 3, height, width))


def gaussian(shape, seed: int) -> torch.Tensor:
    n = int(np.prod(shape))
    return torch.from_numpy(normal_like(n, seed).reshape(shape))


def timesteps(batch: int, seed: int = 2, high: int = 1000) -> torch.Tensor:
    h = _hash_u32(np.arange(batch, dtype=np.uint64), seed * 104729 + 5)
    return torch.from_numpy((h % np.uint32(high)).astype(np.int64))


def input_ids(batch: int, placeholder_id: int, vocab_size: int = 49408, seq_len: int = 77,
              view_placeholder_id: int | None = None) -> torch.Tensor:
    """'a photo of a <obj>'-shaped ids: BOS, 4 template tokens, placeholder(s), EOS padding.
    (The CLIP BPE tokenizer files are absent; training/dataset.py:683-689 would produce this.)"""
    bos, eos = vocab_size - 2, vocab_size - 1
    tmpl = [320 % (vocab_size - 2), 1125 % (vocab_size - 2), 539 % (vocab_size - 2), 320 % (vocab_size - 2)]
    row = [bos] + tmpl
    if view_placeholder_id is not None:
        row.append(view_placeholder_id)
    row.append(placeholder_id)
    row += [eos] * (seq_len - len(row))
    ids = torch.tensor(row, dtype=torch.int64).unsqueeze(0).repeat(batch, 1)
    return ids
