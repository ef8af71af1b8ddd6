"""The eight augmentation pipelines of the reference's dataset (training/dataset.py:238-316) without
torchvision (absent from this image): each transform is restated on PIL images + numpy following
torchvision 0.14's PIL code paths (`transforms.py`, `functional_pil.py`), drawing its random numbers from
the torch global RNG with the same calls in the same order (`torch.rand(1)`, `torch.randperm(4)`,
`torch.empty(1).uniform_`, `torch.randint`), so a seeded run consumes the RNG stream the way the reference does.

    key 1: jitter(.75) gray(.1) blur(.10) rot(.75) rrcrop(.85,1.15)      key 5: jitter(.75) blur(.25) rrcrop(.95,1.05)
    key 2: jitter(.75) gray(.1) blur(.10)                                key 6: jitter gray blur rot rrcrop(.70,1.3)
    key 3: jitter gray blur rot(.75)                                     key 7: jitter blur(.2) rot rrcrop(.70,1.3)
    key 4: jitter gray blur rrcrop(.85,1.15)                             key 8: jitter gray blur(.10)

This is host code of the input pipeline (SURVEY §8 f3 marks the GPU-side version "next"); it exists so the
reference's shipped YAMLs (augmentation_key 5 / 7) run unmodified.
"""
from __future__ import annotations

import math
from typing import Callable, List, Tuple

import numpy as np
import torch
from PIL import Image, ImageEnhance


# ------------------------------------------------------------------ primitives (torchvision functional_pil.py)
def adjust_hue(img: Image.Image, hue_factor: float) -> Image.Image:
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    # uint8 wrap-around is the hue rotation (np.uint8(negative float) wrapped the same way before numpy 2)
    np_h = (np_h.astype(np.int32) + (int(hue_factor * 255) % 256)).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def color_jitter(img: Image.Image, b=0.04, c=0.04, s=0.04, h=0.04) -> Image.Image:
    """T.ColorJitter.forward: random order of the four adjustments, factors drawn b, c, s, h."""
    order = torch.randperm(4)
    fb = float(torch.empty(1).uniform_(max(0.0, 1 - b), 1 + b))
    fc = float(torch.empty(1).uniform_(max(0.0, 1 - c), 1 + c))
    fs = float(torch.empty(1).uniform_(max(0.0, 1 - s), 1 + s))
    fh = float(torch.empty(1).uniform_(-h, h))
    for fn_id in order.tolist():
        if fn_id == 0:
            img = ImageEnhance.Brightness(img).enhance(fb)
        elif fn_id == 1:
            img = ImageEnhance.Contrast(img).enhance(fc)
        elif fn_id == 2:
            img = ImageEnhance.Color(img).enhance(fs)
        else:
            img = adjust_hue(img, fh)
    return img


def to_grayscale3(img: Image.Image) -> Image.Image:
    g = np.array(img.convert("L"), dtype=np.uint8)
    return Image.fromarray(np.dstack([g, g, g]), "RGB")


def gaussian_blur(img: Image.Image, kernel_size: int = 5, sigma_range=(0.1, 0.2)) -> Image.Image:
    """T.GaussianBlur: sigma ~ U(range); separable kernel on the float image with reflect padding, result
    rounded back to uint8 (functional_tensor.gaussian_blur via the PIL->tensor->PIL round trip)."""
    sigma = float(torch.empty(1).uniform_(sigma_range[0], sigma_range[1]))
    half = (kernel_size - 1) * 0.5
    x = np.linspace(-half, half, kernel_size)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    k = (k / k.sum()).astype(np.float32)
    a = np.asarray(img, dtype=np.float32)
    p = kernel_size // 2
    a = np.pad(a, ((p, p), (p, p), (0, 0)), mode="reflect")
    H, W = a.shape[0] - 2 * p, a.shape[1] - 2 * p
    tmp = sum(k[i] * a[:, i:i + W] for i in range(kernel_size))
    out = sum(k[i] * tmp[i:i + H] for i in range(kernel_size))
    return Image.fromarray(np.clip(np.round(out), 0, 255).astype(np.uint8), "RGB")


def random_rotation(img: Image.Image, degrees: float = 10.0, fill: int = 1) -> Image.Image:
    """T.RandomRotation(degrees, fill): nearest resampling, no expansion, constant fill."""
    angle = float(torch.empty(1).uniform_(-degrees, degrees))
    return img.rotate(angle, resample=Image.NEAREST, expand=False, fillcolor=(fill,) * len(img.getbands()))


def random_resized_crop(img: Image.Image, size: Tuple[int, int], scale: Tuple[float, float],
                        ratio=(3.0 / 4.0, 4.0 / 3.0)) -> Image.Image:
    """T.RandomResizedCrop(size=(h, w), scale, ratio) with bilinear resizing."""
    width, height = img.size
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    box = None
    for _ in range(10):
        target_area = area * float(torch.empty(1).uniform_(scale[0], scale[1]))
        aspect = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, size=(1,)))
            j = int(torch.randint(0, width - w + 1, size=(1,)))
            box = (i, j, h, w)
            break
    if box is None:  # fallback: central crop clamped to the ratio range
        in_ratio = float(width) / float(height)
        if in_ratio < min(ratio):
            w, h = width, int(round(width / min(ratio)))
        elif in_ratio > max(ratio):
            h, w = height, int(round(height * max(ratio)))
        else:
            w, h = width, height
        box = ((height - h) // 2, (width - w) // 2, h, w)
    i, j, h, w = box
    return img.crop((j, i, j + w, i + h)).resize((size[1], size[0]), resample=Image.BILINEAR)


# ------------------------------------------------------------------ pipelines
def _maybe(p: float, fn: Callable[[Image.Image], Image.Image]):
    """T.RandomApply([t], p): skipped when p < torch.rand(1)."""
    def run(img):
        if p < float(torch.rand(1)):
            return img
        return fn(img)
    return run


def _gray(p: float):
    """T.RandomGrayscale(p): applied when torch.rand(1) < p."""
    def run(img):
        return to_grayscale3(img) if float(torch.rand(1)) < p else img
    return run


def build_augmentations(key: int, size: Tuple[int, int]) -> Callable[[Image.Image], Image.Image]:
    """size = (height, width) as in dataset.py:229-236."""
    jitter = _maybe(0.75, color_jitter)
    rot = _maybe(0.75, random_rotation)
    blur = lambda p: _maybe(p, gaussian_blur)
    crop = lambda lo, hi: (lambda img: random_resized_crop(img, size, (lo, hi)))
    table = {
        1: [jitter, _gray(0.1), blur(0.10), rot, crop(0.85, 1.15)],
        2: [jitter, _gray(0.1), blur(0.10)],
        3: [jitter, _gray(0.1), blur(0.10), rot],
        4: [jitter, _gray(0.1), blur(0.10), crop(0.85, 1.15)],
        5: [jitter, blur(0.25), crop(0.95, 1.05)],
        6: [jitter, _gray(0.1), blur(0.10), rot, crop(0.70, 1.3)],
        7: [jitter, blur(0.2), rot, crop(0.70, 1.3)],
        8: [jitter, _gray(0.1), blur(0.10)],
    }
    if key not in table:
        raise ValueError(f"unknown augmentation_key {key}")  # the reference does a bare `raise` (dataset.py:315)
    steps: List[Callable] = table[key]

    def run(img: Image.Image) -> Image.Image:
        for t in steps:
            img = t(img)
        return img
    return run
