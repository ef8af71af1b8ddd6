"""The DTU novel-view metric harness of the reference: `training/inference_dtu.py:46-84, 375-645`
and the part of `training/validate.py:65-186` that turns generated views into masked MSE / PSNR / SSIM numbers and
the ground-truth | prediction | masked | residual grids.

Host-side evaluation glue (34 images of 300x400 per run): numpy / scipy / torch-CPU, not a kernel target.  Third-party
routines the reference calls are restated from their published algorithms because the packages are not in this image:

* `skimage.metrics.structural_similarity(x, y, channel_axis=0, data_range=1.0)` (scikit-image 0.19+): 7x7 uniform
  window (scipy `uniform_filter`, reflect mode), sample covariance (N/(N-1)), K1 = 0.01, K2 = 0.03, computed in the
  input's float type, border of 3 pixels cropped, mean in f64, then the mean over channels.  **Parity unpinned**: no
  scikit-image here to generate golden values; the tests pin the restatement against a direct per-window evaluation
  of the SSIM definition and its invariants only.
* `torchvision.transforms.Resize((300, 400), BICUBIC)` on uint8 tensors (torchvision 0.14 tensor path, no
  antialias): f32 `interpolate(mode="bicubic", align_corners=False)`, clamp to [0, 255], round, cast.
* `torchvision.utils.make_grid(t, nrow)`: padding 2, pad value 0.
* LPIPS(net="vgg") needs the VGG16 + linear-head weights of the `lpips` package, which are not available offline:
  `lpips_fn_batch` raises; `do_lpips=False` (the reference's default, inference_dtu.py:481) reports zeros like the
  reference does.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image, ImageOps

from .constants import DTU_SPLIT_IDXS
from .dataset import TextualInversionDataset

DTU_MASKS = "data/dtu/submission_data/idrmasks"  # constants.py:31


# ---------------------------------------------------------------------------------------------- data collection
def get_cam_idxs(dtu_subset):
    """inference_dtu.py:46-56: all evaluation views of the split (sorted), the training views of the subset, the rest"""
    cam_idxs = sorted(DTU_SPLIT_IDXS["train"] + DTU_SPLIT_IDXS["test"])
    cam_idxs_train = TextualInversionDataset.dtu_get_train_idxs(dtu_subset)
    cam_idxs_test = [i for i in cam_idxs if i not in cam_idxs_train]
    return cam_idxs, cam_idxs_train, cam_idxs_test


def dtu_get_gt_images(cam_idxs, train_data_dir, dtu_lighting, dtu_preprocess_key) -> Dict[int, Image.Image]:
    """inference_dtu.py:59-84 (key 0: pad 400 black rows, bicubic to 768^2; key 1: PIL's default-filter resize to
    768x576)"""
    out = {}
    for idx in cam_idxs:
        f = Path(train_data_dir) / TextualInversionDataset.dtu_cam_and_lighting_to_fname(idx, dtu_lighting)
        image = Image.open(f)
        if dtu_preprocess_key == 0:
            image = ImageOps.expand(image, (0, 0, 0, 400), fill="black")
            assert image.size == (1600, 1600)
            image = image.resize((768, 768), resample=Image.BICUBIC)
        elif dtu_preprocess_key == 1:
            image = image.resize((768, 576))
        else:
            raise NotImplementedError
        out[idx] = image
    return out


def get_object_masks(cam_idxs, scan_idx, dtu_preprocess_key=1, masks_root=DTU_MASKS) -> Dict[int, Image.Image]:
    """inference_dtu.py:375-398: the IDR object masks (either `<root>/scanN/mask/NNN.png` or `<root>/scanN/NNN.png`);
    a view without a mask file counts as fully foreground"""
    scan_dir = Path(masks_root) / f"scan{scan_idx}"
    folder = scan_dir / "mask" if (scan_dir / "mask").exists() else scan_dir
    masks = {}
    for cam in cam_idxs:
        f = folder / f"{cam:03d}.png"
        m = Image.open(f).convert("RGB") if f.exists() else Image.new("RGB", (1600, 1200), (255, 255, 255))
        masks[cam] = m.resize((400, 300)) if dtu_preprocess_key == 1 else m
    return masks


# ---------------------------------------------------------------------------------------------- tensor helpers
def resize_bicubic_uint8(t: torch.Tensor, size) -> torch.Tensor:
    """T.Resize(size, BICUBIC) on a uint8 (..., C, H, W) tensor (tensor path, antialias off)"""
    assert t.dtype == torch.uint8 and t.dim() == 4
    x = F.interpolate(t.to(torch.float32), size=tuple(size), mode="bicubic", align_corners=False)
    return torch.round(x.clamp(min=0, max=255)).to(torch.uint8)


def make_grid(t: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """torchvision.utils.make_grid for a (B, C, H, W) tensor"""
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.size(0) == 1:
        return t.squeeze(0)
    nmaps = t.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(np.ceil(float(nmaps) / xmaps))
    height, width = int(t.size(2) + padding), int(t.size(3) + padding)
    grid = t.new_full((t.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid[:, y * height + padding:y * height + padding + t.size(2),
                 x * width + padding:x * width + padding + t.size(3)] = t[k]
            k += 1
    return grid


def _stack_chw(lookup, cam_idxs) -> torch.Tensor:
    """dict camidx -> HWC uint8 image(s)  ->  uint8 tensor with channels before the spatial axes"""
    arr = np.stack([np.asarray(lookup[c]) for c in cam_idxs])
    return torch.from_numpy(arr).movedim(-1, -3)


def process_imgs(cam_idxs, cam_idxs_train, lookup_camidx_to_img_pred, lookup_camidx_to_img_gt, lookup_camidx_to_mask):
    """inference_dtu.py:401-465: predictions (views, seeds, C, H, W), ground truth and masks (views, C, H, W), all
    brought to the 300 x 400 evaluation size in [0, 1]; masks binarised at 0.01; plus the ground truth with a 50-row
    header (yellow on training views) for the figure.  Returns (pred, gt, masks, gt, gt_plot) like the reference."""
    size = (300, 400)
    pred = _stack_chw(lookup_camidx_to_img_pred, cam_idxs)
    gt = _stack_chw(lookup_camidx_to_img_gt, cam_idxs)
    mk = _stack_chw(lookup_camidx_to_mask, cam_idxs)
    if pred.dim() != 5 or gt.dim() != 4 or mk.dim() != 4:
        raise ValueError("expected predictions (views, seeds, h, w, 3) and ground truth / masks (views, h, w, 3)")
    for t in (pred, gt):
        if t.shape[-2] / t.shape[-1] != 0.75:
            raise ValueError("the DTU evaluation assumes 3:4 images")
    n_views, n_seeds = pred.shape[:2]
    gt = resize_bicubic_uint8(gt, size)
    mk = resize_bicubic_uint8(mk, size)
    pred = resize_bicubic_uint8(pred.flatten(0, 1), size).unflatten(0, (n_views, n_seeds))
    header = torch.zeros(n_views, 3, 50, size[1])
    is_train = torch.tensor([c in cam_idxs_train for c in cam_idxs])
    header[is_train] = torch.tensor([255.0, 255.0, 0.0]).view(1, 3, 1, 1)
    gt_plot = torch.cat((header, gt.float()), dim=2) / 255.0
    pred, gt = pred / 255.0, gt / 255.0
    masks = (mk / 255.0 > 0.01).to(gt.dtype)
    return pred, gt, masks, gt, gt_plot


# ---------------------------------------------------------------------------------------------- metrics
def mse_to_psnr(mse):
    """inference_dtu.py:606-613 (peak value 1)"""
    return -10.0 / np.log(10.0) * np.log(mse)


def structural_similarity(im1: np.ndarray, im2: np.ndarray, *, win_size: int = 7, data_range: float = 1.0,
                          channel_axis=None) -> float:
    """skimage.metrics.structural_similarity with its defaults (uniform window, sample covariance)"""
    from scipy.ndimage import uniform_filter
    if channel_axis is not None:
        im1, im2 = np.moveaxis(im1, channel_axis, -1), np.moveaxis(im2, channel_axis, -1)
        per = [structural_similarity(im1[..., c], im2[..., c], win_size=win_size, data_range=data_range)
               for c in range(im1.shape[-1])]
        return float(np.asarray(per, dtype=np.float64).mean())
    if min(im1.shape) < win_size:
        raise ValueError("win_size exceeds image extent")
    ftype = np.float32 if im1.dtype in (np.float16, np.float32) else np.float64
    im1, im2 = im1.astype(ftype, copy=False), im2.astype(ftype, copy=False)
    K1, K2 = 0.01, 0.03
    NP = win_size ** im1.ndim
    cov_norm = NP / (NP - 1)
    filt = lambda a: uniform_filter(a, size=win_size)
    ux, uy = filt(im1), filt(im2)
    uxx, uyy, uxy = filt(im1 * im1), filt(im2 * im2), filt(im1 * im2)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
    S = (A1 * A2) / (B1 * B2)
    pad = (win_size - 1) // 2
    core = S[tuple(slice(pad, -pad) for _ in range(S.ndim))]
    return float(core.mean(dtype=np.float64))


def ssim_fn(x, y):
    assert x.ndim == 3
    return structural_similarity(x, y, channel_axis=0, data_range=1.0)


def ssim_fn_batch(x, y) -> torch.Tensor:
    x, y = np.asarray(x), np.asarray(y)
    return torch.tensor([ssim_fn(a, b) for a, b in zip(x, y)])


def mse_batch(imgs_gt: torch.Tensor, imgs_pred: torch.Tensor) -> torch.Tensor:
    bs = len(imgs_gt)
    return ((imgs_gt - imgs_pred) ** 2).view(bs, -1).mean(1)


def lpips_fn_batch(imgs_gt, imgs_pred, lpips_fn=None):
    if lpips_fn is None:
        raise NotImplementedError("LPIPS(net='vgg') needs the lpips package's VGG16 + linear-head weights, which are "
                                  "not available offline; pass a callable `lpips_fn(pred, gt) -> (B,1,1,1)`")
    assert imgs_gt.min() >= 0 and imgs_gt.max() <= 1
    with torch.no_grad():
        return lpips_fn(imgs_pred * 2 - 1, imgs_gt * 2 - 1)[:, 0, 0, 0].cpu()


def get_result_metrics_and_grids(cam_idxs, cam_idxs_train, imgs_pred_all_seeds, imgs_gt, masks, imgs_gt_plot, seeds,
                                 do_lpips: bool = False, title_prefix: str = "", lpips_fn=None,
                                 make_figures: bool = True) -> dict:
    """inference_dtu.py:468-604: per-seed masked metrics, train/test means, and the 4-row grids"""
    is_train = torch.tensor([idx in cam_idxs_train for idx in cam_idxs])
    acc = {k: ([], []) for k in ("mse", "psnr", "ssim", "lpips")}
    grids, figures, all_pred = [], [], []
    for si, seed in enumerate(seeds):
        imgs_pred = imgs_pred_all_seeds[:, si]
        all_pred.append(imgs_pred)
        bs = len(imgs_pred)
        mse_b = (((imgs_gt * masks) - (imgs_pred * masks)) ** 2).reshape(bs, -1).sum(dim=1) / masks.reshape(bs, -1).sum(dim=1)
        psnr_b = mse_to_psnr(mse_b)
        ssim_b = ssim_fn_batch(imgs_pred * masks, imgs_gt * masks)
        lpips_b = lpips_fn_batch(imgs_pred * masks, imgs_gt * masks, lpips_fn) if do_lpips else torch.zeros_like(ssim_b)
        for name, v in (("mse", mse_b), ("psnr", psnr_b), ("ssim", ssim_b), ("lpips", lpips_b)):
            acc[name][0].append(v[is_train])
            acc[name][1].append(v[~is_train])
        residual = ((imgs_pred - imgs_gt) + 1) / 2
        nrow = len(imgs_gt)
        grid = torch.cat((make_grid(imgs_gt_plot, nrow=nrow), make_grid(imgs_pred, nrow=nrow),
                          make_grid(imgs_pred * masks, nrow=nrow), make_grid(residual, nrow=nrow)), dim=1)
        grid = grid.permute(1, 2, 0)
        grids.append(grid)
        if make_figures:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            m = lambda v, sel: v[sel].mean().item()
            title = (title_prefix + f" PSNR: train {m(psnr_b, is_train):.3f}   test {m(psnr_b, ~is_train):.3f}  |  "
                     f"MSE: train {m(mse_b, is_train):.3f}   test {m(mse_b, ~is_train):.3f}  |  "
                     f"SSIM: train {m(ssim_b, is_train):.3f}   test {m(ssim_b, ~is_train):.3f}  |  "
                     f"LPIPS: train {m(lpips_b, is_train):.3f}   test {m(lpips_b, ~is_train):.3f}  |  ")
            labels = []
            for i, (tr, p, ms, s, l) in enumerate(zip(is_train, psnr_b, mse_b, ssim_b, lpips_b)):
                label = f"{p:.1f}\n{ms:.4f}\n{s:.3f}\n{l:.3f}"
                if i == 0:
                    label = "\n".join(a + b for a, b in zip(["psnr ", "mse ", "ssim ", "lpips"], label.split("\n")))
                if tr:
                    label += "\nTRAIN"
                labels.append(label)
            ydim = imgs_gt.shape[2]
            xticks = np.linspace(0, grid.shape[1] - ydim, len(labels)) + ydim // 2
            f, axs = plt.subplots(figsize=(nrow, 5))
            axs.imshow(grid.clamp(0, 1).numpy())
            axs.set_xticks(xticks)
            axs.set_xticklabels(labels, fontsize=6)
            axs.set_yticks([])
            axs.set(title=title)
            figures.append(f)
    out = dict(figures=figures, grids=grids, imgs_pred=all_pred, imgs_gt=imgs_gt, imgs_gt_plot=imgs_gt_plot, masks=masks)
    for name, (tr, te) in acc.items():
        out[f"{name}_train_mean"] = torch.cat(tr).mean().item()
        out[f"{name}_test_mean"] = torch.cat(te).mean().item()
    return out


def evaluate_dtu_predictions(lookup_camidx_to_img_pred: Dict[int, np.ndarray], train_data_dir, dtu_subset, dtu_lighting,
                             dtu_preprocess_key, seeds: Sequence[int], scan_id=None, masks_root=DTU_MASKS,
                             do_lpips: bool = False, make_figures: bool = True) -> dict:
    """The tail of ValidationHandler.infer_dtu (validate.py:123-152): predictions (camidx -> (n_seeds, H, W, 3) uint8)
    against the scene's ground-truth views and object masks."""
    cam_idxs, cam_idxs_train, _ = get_cam_idxs(dtu_subset)
    assert set(lookup_camidx_to_img_pred.keys()) == set(cam_idxs)
    if scan_id is None:
        scan_id = Path(train_data_dir).stem[4:]
    gt = dtu_get_gt_images(cam_idxs, train_data_dir, dtu_lighting, dtu_preprocess_key)
    masks = get_object_masks(cam_idxs, scan_id, masks_root=masks_root)
    pred, gt_t, masks_t, _, gt_plot = process_imgs(cam_idxs, cam_idxs_train, lookup_camidx_to_img_pred, gt, masks)
    return get_result_metrics_and_grids(cam_idxs, cam_idxs_train, pred, gt_t, masks_t, gt_plot, list(seeds),
                                        do_lpips=do_lpips, make_figures=make_figures)
