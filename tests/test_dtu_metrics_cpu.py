"""the DTU metric harness (compat/dtu_metrics.py, restating training/inference_dtu.py:401-645).
scikit-image / torchvision / lpips are not installable here, so the third-party pieces are pinned against direct
evaluations of their published definitions (SSIM per window, make_grid layout) and end-to-end invariants."""
import numpy as np
import pytest
import torch
from PIL import Image


def _ssim_direct(x, y, win=7, R=1.0):
    """the SSIM definition evaluated window by window (uniform 7x7, sample covariance), valid positions only"""
    K1, K2 = 0.01, 0.03
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    H, W = x.shape
    vals = []
    for i in range(H - win + 1):
        for j in range(W - win + 1):
            a, b = x[i:i + win, j:j + win].astype(np.float64), y[i:i + win, j:j + win].astype(np.float64)
            ux, uy = a.mean(), b.mean()
            vx, vy = a.var(ddof=1), b.var(ddof=1)
            vxy = ((a - ux) * (b - uy)).sum() / (win * win - 1)
            vals.append((2 * ux * uy + C1) * (2 * vxy + C2) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2)))
    return float(np.mean(vals))


def test_ssim_matches_definition_and_invariants():
    from view_neti_amd.compat.dtu_metrics import ssim_fn, ssim_fn_batch, structural_similarity
    rng = np.random.default_rng(0)
    x = rng.random((24, 31)).astype(np.float64)
    y = np.clip(x + 0.1 * rng.standard_normal(x.shape), 0, 1)
    assert abs(structural_similarity(x, y) - _ssim_direct(x, y)) < 1e-10
    x3 = rng.random((3, 20, 26)).astype(np.float32)
    y3 = np.clip(x3 + 0.05 * rng.standard_normal(x3.shape).astype(np.float32), 0, 1)
    ref = np.mean([_ssim_direct(x3[c], y3[c]) for c in range(3)])
    assert abs(ssim_fn(x3, y3) - ref) < 5e-5  # f32 input: computed in f32 like scikit-image does
    assert abs(ssim_fn(x3, x3) - 1.0) < 1e-6 and abs(ssim_fn(x3, y3) - ssim_fn(y3, x3)) < 1e-6
    b = ssim_fn_batch(torch.from_numpy(np.stack([x3, y3])), torch.from_numpy(np.stack([y3, y3])))
    assert b.shape == (2,) and abs(b[1].item() - 1.0) < 1e-6
    with pytest.raises(ValueError):
        structural_similarity(x[:5, :5], y[:5, :5])


def test_resize_grid_and_psnr_helpers():
    from view_neti_amd.compat import dtu_metrics as dm
    t = torch.full((2, 3, 12, 16), 77, dtype=torch.uint8)
    r = dm.resize_bicubic_uint8(t, (30, 40))
    assert r.dtype == torch.uint8 and r.shape == (2, 3, 30, 40) and int(r.min()) == int(r.max()) == 77
    ramp = torch.arange(16, dtype=torch.uint8).repeat(1, 1, 12, 1) * 10
    rr = dm.resize_bicubic_uint8(ramp, (24, 32)).float()
    assert (rr[0, 0, 0, 1:] - rr[0, 0, 0, :-1] >= -1).all()  # monotone up to rounding
    g = dm.make_grid(torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).view(5, 3, 4, 6), nrow=3)
    assert g.shape == (3, 2 * 6 + 2, 3 * 8 + 2) and g[0, 0, 0] == 0 and g[0, 2, 2] == 0.0 + 0
    assert torch.equal(g[:, 2:6, 10:16], torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).view(5, 3, 4, 6)[1])
    assert abs(dm.mse_to_psnr(0.01) - 20.0) < 1e-9
    with pytest.raises(NotImplementedError):
        dm.lpips_fn_batch(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))


def test_evaluate_dtu_predictions_end_to_end(tmp_path):
    """synthetic scene: 34 evaluation views, lighting 3; predictions = ground truth (+ noise on the test views): the harness
    must report ~perfect train metrics, worse test metrics, honour the masks, and lay the grid out as the reference"""
    from view_neti_amd.compat import dtu_metrics as dm
    from view_neti_amd.compat.dataset import TextualInversionDataset as DS
    scene = tmp_path / "scan114"
    scene.mkdir()
    rng = np.random.default_rng(1)
    cam_idxs, cam_train, cam_test = dm.get_cam_idxs(3)
    n = len(cam_idxs)  # the pixelNeRF split: 9 'train' + 25 'test' views
    assert n == 34 and len(cam_train) == 3 and len(cam_test) == n - 3
    for c in cam_idxs:
        low = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
        Image.fromarray(low).resize((160, 120), Image.BICUBIC).save(scene / DS.dtu_cam_and_lighting_to_fname(c, "3"))
    masks_root = tmp_path / "masks"
    (masks_root / "scan114" / "mask").mkdir(parents=True)
    m = np.zeros((1200, 1600, 3), np.uint8)
    m[300:900, 400:1200] = 255
    Image.fromarray(m).save(masks_root / "scan114" / "mask" / f"{cam_idxs[0]:03d}.png")  # the others: all-white
    gt = dm.dtu_get_gt_images(cam_idxs, scene, "3", 1)
    assert gt[cam_idxs[0]].size == (768, 576)
    pred = {}
    for c in cam_idxs:
        a = np.asarray(gt[c]).astype(np.int32)
        noisy = np.clip(a + rng.integers(-40, 41, a.shape), 0, 255)
        pred[c] = np.stack([a if c in cam_train else noisy, noisy]).astype(np.uint8)  # seed 0 exact on train views
    res = dm.evaluate_dtu_predictions(pred, scene, 3, "3", 1, seeds=[0, 1], masks_root=str(masks_root),
                                      make_figures=True)
    assert res["masks"].shape == (n, 3, 300, 400) and set(res["masks"].unique().tolist()) == {0.0, 1.0}
    assert 0.2 < res["masks"][0].mean().item() < 0.3 and res["masks"][1].mean().item() == 1.0
    assert res["mse_train_mean"] < res["mse_test_mean"] and res["psnr_train_mean"] > res["psnr_test_mean"]
    assert res["ssim_train_mean"] > res["ssim_test_mean"] and res["lpips_test_mean"] == 0.0
    # seed 0 is exact on the training views: per-view metrics there are perfect
    p0, g0, mk = res["imgs_pred"][0], res["imgs_gt"], res["masks"]
    tr = torch.tensor([c in cam_train for c in cam_idxs])
    assert torch.equal(p0[tr], g0[tr])
    assert abs(dm.ssim_fn_batch(p0[tr] * mk[tr], g0[tr] * mk[tr]).mean().item() - 1.0) < 1e-6
    grid = res["grids"][0]
    assert grid.shape == (354 + 3 * 304, 2 + n * 402, 3) and len(res["figures"]) == 2
    assert torch.allclose(grid[2:52, 2 + 402 * cam_idxs.index(cam_train[0]):2 + 402 * cam_idxs.index(cam_train[0]) + 400],
                          torch.tensor([1.0, 1.0, 0.0]).expand(50, 400, 3))  # yellow header on a training view
