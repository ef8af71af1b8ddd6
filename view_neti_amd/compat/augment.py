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


def blur_kernel(sigma: float, kernel_size: int = 5) -> np.ndarray:
    half = (kernel_size - 1) * 0.5
    x = np.linspace(-half, half, kernel_size)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    return (k / k.sum()).astype(np.float32)


def gaussian_blur(img: Image.Image, kernel_size: int = 5, sigma_range=(0.1, 0.2)) -> Image.Image:
    """T.GaussianBlur: sigma ~ U(range); separable kernel on the float image with reflect padding, result
    rounded back to uint8 (functional_tensor.gaussian_blur via the PIL->tensor->PIL round trip)."""
    return blur_with_sigma(img, float(torch.empty(1).uniform_(sigma_range[0], sigma_range[1])), kernel_size)


def blur_with_sigma(img: Image.Image, sigma: float, kernel_size: int = 5) -> Image.Image:
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


# ------------------------------------------------------------------ pipelines: draw a plan, then apply it
# A pipeline run is split in two: `draw_plan` consumes the torch RNG exactly like the torchvision transforms do and
# returns the concrete operations with their parameters; `apply_plan` executes them on a PIL image.  The device
# input pipeline (engine/input_pipeline.py) executes the SAME plan with HIP kernels, so a seeded run
# is identical on both paths by construction of the draws and bit-comparable by the kernels' parity tests.
def _draw_jitter(b=0.04, c=0.04, s=0.04, h=0.04):
    order = torch.randperm(4).tolist()
    fb = float(torch.empty(1).uniform_(max(0.0, 1 - b), 1 + b))
    fc = float(torch.empty(1).uniform_(max(0.0, 1 - c), 1 + c))
    fs = float(torch.empty(1).uniform_(max(0.0, 1 - s), 1 + s))
    fh = float(torch.empty(1).uniform_(-h, h))
    return ("jitter", order, fb, fc, fs, fh)


def _draw_blur(sigma_range=(0.1, 0.2)):
    return ("blur", float(torch.empty(1).uniform_(sigma_range[0], sigma_range[1])))


def _draw_rotation(degrees: float = 10.0):
    return ("rotate", float(torch.empty(1).uniform_(-degrees, degrees)))


def _draw_rrcrop(width: int, height: int, size: Tuple[int, int], scale, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * float(torch.empty(1).uniform_(scale[0], scale[1]))
        aspect = math.exp(float(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = int(torch.randint(0, height - h + 1, size=(1,)))
            j = int(torch.randint(0, width - w + 1, size=(1,)))
            return ("rrcrop", i, j, h, w, size[0], size[1])
    in_ratio = float(width) / float(height)  # fallback: central crop clamped to the ratio range
    if in_ratio < min(ratio):
        w, h = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        h, w = height, int(round(height * max(ratio)))
    else:
        w, h = width, height
    return ("rrcrop", (height - h) // 2, (width - w) // 2, h, w, size[0], size[1])


# (kind, probability, arguments) per step; RandomApply skips when p < rand, RandomGrayscale applies when rand < p
_PIPELINES = {
    1: [("jitter", 0.75), ("gray", 0.1), ("blur", 0.10), ("rotate", 0.75), ("rrcrop", (0.85, 1.15))],
    2: [("jitter", 0.75), ("gray", 0.1), ("blur", 0.10)],
    3: [("jitter", 0.75), ("gray", 0.1), ("blur", 0.10), ("rotate", 0.75)],
    4: [("jitter", 0.75), ("gray", 0.1), ("blur", 0.10), ("rrcrop", (0.85, 1.15))],
    5: [("jitter", 0.75), ("blur", 0.25), ("rrcrop", (0.95, 1.05))],
    6: [("jitter", 0.75), ("gray", 0.1), ("blur", 0.10), ("rotate", 0.75), ("rrcrop", (0.70, 1.3))],
    7: [("jitter", 0.75), ("blur", 0.2), ("rotate", 0.75), ("rrcrop", (0.70, 1.3))],
    8: [("jitter", 0.75), ("gray", 0.1), ("blur", 0.10)],
}


def draw_plan(key: int, size: Tuple[int, int], width: int, height: int) -> List[tuple]:
    """The random draws of one pipeline run on a width x height image (size = (height, width) target of the crops,
    dataset.py:229-236), in torchvision's order."""
    if key not in _PIPELINES:
        raise ValueError(f"unknown augmentation_key {key}")  # the reference does a bare `raise` (dataset.py:315)
    plan: List[tuple] = []
    for kind, arg in _PIPELINES[key]:
        if kind == "gray":
            if float(torch.rand(1)) < arg:
                plan.append(("gray",))
        elif kind == "rrcrop":
            op = _draw_rrcrop(width, height, size, arg)
            plan.append(op)
            height, width = op[5], op[6]
        else:
            if arg < float(torch.rand(1)):  # T.RandomApply([t], p)
                continue
            plan.append({"jitter": _draw_jitter, "blur": _draw_blur, "rotate": _draw_rotation}[kind]())
    return plan


def apply_plan(img: Image.Image, plan: List[tuple], fill: int = 1) -> Image.Image:
    for op in plan:
        kind = op[0]
        if kind == "jitter":
            _, order, fb, fc, fs, fh = op
            for fn_id in order:
                if fn_id == 0:
                    img = ImageEnhance.Brightness(img).enhance(fb)
                elif fn_id == 1:
                    img = ImageEnhance.Contrast(img).enhance(fc)
                elif fn_id == 2:
                    img = ImageEnhance.Color(img).enhance(fs)
                else:
                    img = adjust_hue(img, fh)
        elif kind == "gray":
            img = to_grayscale3(img)
        elif kind == "blur":
            img = blur_with_sigma(img, op[1])
        elif kind == "rotate":
            img = img.rotate(op[1], resample=Image.NEAREST, expand=False, fillcolor=(fill,) * len(img.getbands()))
        elif kind == "rrcrop":
            _, i, j, h, w, oh, ow = op
            img = img.crop((j, i, j + w, i + h)).resize((ow, oh), resample=Image.BILINEAR)
        else:
            raise ValueError(kind)
    return img


def build_augmentations(key: int, size: Tuple[int, int]) -> Callable[[Image.Image], Image.Image]:
    """size = (height, width) as in dataset.py:229-236."""
    if key not in _PIPELINES:
        raise ValueError(f"unknown augmentation_key {key}")

    def run(img: Image.Image) -> Image.Image:
        return apply_plan(img, draw_plan(key, size, img.size[0], img.size[1]))
    return run
