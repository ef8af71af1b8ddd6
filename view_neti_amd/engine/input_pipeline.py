"""Device-side input pipeline: the per-sample image work of the reference's dataset
(training/dataset.py:238-316 augmentation pipelines, :700-740 resize / flip / normalise) as HIP kernels over uint8
images cached in HBM, writing the normalised f32 CHW planes straight into the train step's pixel buffer.

Division of labour: the HOST draws the random parameters (`compat/augment.py::draw_plan`, torchvision's RNG calls in
torchvision's order, so a seeded run consumes the generator exactly like the reference) and the DEVICE executes the
plan (`csrc/image.hip`, each kernel restating the Pillow / torchvision arithmetic).  The host path
(`compat/augment.py::apply_plan`, PIL) executes the same plan; `tests/test_input_pipeline_gpu.py` compares the two.

    pipe = DeviceImagePipeline(max_h, max_w)
    src = pipe.upload(np_uint8_hwc)                              # once per training image
    pipe.run(src, out=engine.pixel_values[b], resize=(h, w), flip=False, plan=draw_plan(key, (h, w), w, h))
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import ops

BICUBIC, BILINEAR = 0, 1


def rotation_coefficients(angle: float, w: int, h: int) -> List[int]:
    """Image.rotate(angle, expand=False) -> the affine matrix of Image.py::rotate -> Geometry.c affine_fixed's six
    16.16 integers (FIX(v) = floor(v * 65536 + 0.5), half-pixel centre folded into a2 / a5)."""
    angle = angle % 360.0
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5),
            fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


class DeviceImagePipeline:
    def __init__(self, max_h: int, max_w: int, device: str = "cuda"):
        self.dev = device
        self.max_h, self.max_w = max_h, max_w
        n = max_h * max_w * 3
        self._a = torch.empty(n, dtype=torch.uint8, device=device)   # ping-pong images
        self._b = torch.empty(n, dtype=torch.uint8, device=device)
        self._tmpf = torch.empty(n, dtype=torch.float32, device=device)
        self._scratch = torch.zeros(2, dtype=torch.int64, device=device)
        self._coef = {}    # (in, out, filter) -> (bounds, kk, ksize), computed on the device once per size pair
        self._mid = {}

    # ------------------------------------------------------------------ helpers
    def upload(self, arr: np.ndarray) -> Tuple[torch.Tensor, int, int]:
        """cache one source image (uint8 HWC RGB, as dataset.py hands it to `_resize`) in HBM"""
        assert arr.dtype == np.uint8 and arr.ndim == 3 and arr.shape[2] == 3
        assert arr.shape[0] * arr.shape[1] <= 64 * self.max_h * self.max_w, "source image unreasonably large"
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)
        return t, arr.shape[0], arr.shape[1]

    def _coeffs(self, n_in: int, n_out: int, filt: int):
        key = (n_in, n_out, filt)
        if key not in self._coef:
            ks = ops.img_resample_ksize(n_in, n_out, filt)
            bounds = torch.empty(2 * n_out, dtype=torch.int32, device=self.dev)
            kk = torch.empty(n_out * ks, dtype=torch.int32, device=self.dev)
            ops.img_resample_coeffs(n_in, n_out, filt, bounds, kk)
            if len(self._coef) > 4096:
                self._coef.clear()
            self._coef[key] = (bounds, kk, ks)
        return self._coef[key]

    def _resize(self, src, h, w, dst, oh, ow, filt):
        """ImagingResample: horizontal pass into an (h x ow) uint8 intermediate, then the vertical pass; a pass whose
        size does not change is skipped (Resample.c need_horizontal / need_vertical)."""
        if (h, w) == (oh, ow):
            dst[: h * w * 3].copy_(src.reshape(-1)[: h * w * 3])
            return
        cur, cw = src, w
        if ow != w:
            bounds, kk, ks = self._coeffs(w, ow, filt)
            mid = dst if oh == h else self._mid_buf(h * ow * 3)
            ops.img_resample_pass(cur, w, mid, h, ow, bounds, kk, ks, True)
            cur, cw = mid, ow
        if oh != h:
            bounds, kk, ks = self._coeffs(h, oh, filt)
            ops.img_resample_pass(cur, cw, dst, oh, ow, bounds, kk, ks, False)

    def _mid_buf(self, n):
        t = self._mid.get("m")
        if t is None or t.numel() < n:
            t = torch.empty(max(n, self.max_h * self.max_w * 3), dtype=torch.uint8, device=self.dev)
            self._mid["m"] = t
        return t

    # ------------------------------------------------------------------ one sample
    def run(self, src: Tuple[torch.Tensor, int, int], out: Optional[torch.Tensor], resize: Optional[Tuple[int, int]],
            flip: bool = False, plan: Optional[List[tuple]] = None, fill: int = 1) -> Tuple[torch.Tensor, int, int]:
        """src image -> (bicubic resize to `resize` = (h, w)) -> (horizontal flip) -> plan -> out (f32 [3, h, w]).
        Returns the final uint8 image view and its size (for tests)."""
        img, h, w = src
        a, b = self._a, self._b
        if resize is not None and (h, w) != tuple(resize):
            self._resize(img, h, w, a, resize[0], resize[1], BICUBIC)
            h, w = resize
        else:
            a[: h * w * 3].copy_(img.reshape(-1))
        if flip:
            ops.img_crop(a, w, 0, 0, b, h, w, flip=True)
            a, b = b, a
        for op in plan or ():
            kind = op[0]
            if kind == "jitter":
                _, order, fb, fc, fs, fh = op
                for fn_id in order:
                    if fn_id == 0:
                        ops.img_enhance(a, h, w, 0, fb, self._scratch)
                    elif fn_id == 1:
                        ops.img_enhance(a, h, w, 1, fc, self._scratch)
                    elif fn_id == 2:
                        ops.img_enhance(a, h, w, 2, fs, self._scratch)
                    else:
                        ops.img_hue(a, h, w, int(fh * 255) % 256)
            elif kind == "gray":
                ops.img_enhance(a, h, w, 3, 0.0, self._scratch)
            elif kind == "blur":
                from ..compat.augment import blur_kernel
                ops.img_blur5(a, b, self._tmpf, h, w, blur_kernel(op[1]))
                a, b = b, a
            elif kind == "rotate":
                angle = op[1] % 360.0
                if angle != 0.0:  # Image.rotate returns a copy for angle 0
                    # (the 90/180/270 transposes of Image.rotate cannot be hit by a continuous U(-10, 10) draw)
                    ops.img_affine_nearest(a, b, h, w, rotation_coefficients(op[1], w, h), fill)
                    a, b = b, a
            elif kind == "rrcrop":
                _, i, j, ch, cw, oh, ow = op
                if (i, j, ch, cw) != (0, 0, h, w):
                    ops.img_crop(a, w, i, j, b, ch, cw, flip=False)
                    a, b = b, a
                self._resize(a, ch, cw, b, oh, ow, BILINEAR)
                a, b = b, a
                h, w = oh, ow
            else:
                raise ValueError(kind)
        if out is not None:
            assert tuple(out.shape) == (3, h, w) and out.dtype == torch.float32 and out.is_contiguous()
            ops.img_to_f32_chw(a, out, h, w)
        return a[: h * w * 3].view(h, w, 3), h, w
