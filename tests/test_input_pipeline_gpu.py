"""Device-side input pipeline (csrc/image.hip + engine/input_pipeline.py) against the host path it
replaces: Pillow itself for the resizes / enhancers / rotation, compat/augment.py (the torchvision restatement) for the
whole pipelines.  Same plan (same random draws) on both sides; integer kernels are compared bit-exactly, the ones that
go through float conversions (hue) within 1 LSB."""
import numpy as np
import pytest
import torch
from PIL import Image, ImageEnhance

pytestmark = pytest.mark.gpu


def _img(h, w, seed=0, smooth=True):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if smooth:  # natural-image-like: low-pass the noise so resampling has something to interpolate
        f = np.asarray(Image.fromarray(a).resize((w // 8 + 1, h // 8 + 1), Image.BILINEAR).resize((w, h), Image.BICUBIC))
        a = ((f.astype(np.int32) + a // 8) % 256).astype(np.uint8)
    return a


def _pipe(h=576, w=768):
    from view_neti_amd.engine.input_pipeline import DeviceImagePipeline
    return DeviceImagePipeline(h, w)


def _diff(got, ref):
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    return int(d.max()), float((d > 0).mean())


@pytest.mark.parametrize("src,dst", [((600, 800), (512, 512)), ((1200, 1600), (384, 512)), ((300, 400), (576, 768)),
                                     ((512, 512), (512, 512)), ((512, 640), (512, 512))])
def test_bicubic_resize_matches_pillow(src, dst):
    """dataset.py `_resize`: Image.resize(BICUBIC) = Resample.c in 8-bit fixed point, bit exact"""
    pipe = _pipe()
    a = _img(*src, seed=1)
    ref = np.asarray(Image.fromarray(a).resize((dst[1], dst[0]), resample=Image.BICUBIC))
    got, h, w = pipe.run(pipe.upload(a), None, resize=dst)
    torch.cuda.synchronize()
    assert (h, w) == dst
    mx, frac = _diff(got.cpu().numpy(), ref)
    assert mx == 0, f"bicubic {src}->{dst}: max diff {mx}, {frac:.4%} pixels differ"


def test_flip_and_normalise():
    pipe = _pipe()
    a = _img(96, 128, seed=2)
    out = torch.zeros(3, 96, 128, device="cuda")
    pipe.run(pipe.upload(a), out, resize=None, flip=True)
    ref = (np.asarray(Image.fromarray(a).transpose(Image.FLIP_LEFT_RIGHT)).astype(np.uint8) / 127.5 - 1.0).astype(np.float32)
    assert torch.equal(out.cpu(), torch.from_numpy(ref).permute(2, 0, 1))


@pytest.mark.parametrize("factor", [0.96, 1.0, 1.04, 0.3, 1.7])
def test_enhancers_match_pillow(factor):
    """ImageEnhance.Brightness / Contrast / Color = Blend.c with a black / mean-gray / luma degenerate: bit exact"""
    pipe = _pipe()
    a = _img(120, 160, seed=3)
    for mode, enh in ((0, ImageEnhance.Brightness), (1, ImageEnhance.Contrast), (2, ImageEnhance.Color)):
        ref = np.asarray(enh(Image.fromarray(a)).enhance(factor))
        plan = [("jitter", [mode], factor, factor, factor, 0.0)]
        got, _, _ = pipe.run(pipe.upload(a), None, resize=None, plan=plan)
        torch.cuda.synchronize()
        mx, frac = _diff(got.cpu().numpy(), ref)
        assert mx == 0, f"enhance mode {mode} factor {factor}: max diff {mx}, {frac:.4%}"


@pytest.mark.parametrize("fh", [-0.04, -0.013, 0.0, 0.021, 0.04, 0.3])
def test_hue_matches_host(fh):
    from view_neti_amd.compat.augment import adjust_hue, to_grayscale3
    pipe = _pipe()
    a = _img(120, 160, seed=4)
    ref = np.asarray(adjust_hue(Image.fromarray(a), fh))
    got, _, _ = pipe.run(pipe.upload(a), None, resize=None, plan=[("jitter", [3], 1.0, 1.0, 1.0, fh)])
    torch.cuda.synchronize()
    mx, frac = _diff(got.cpu().numpy(), ref)
    assert mx <= 1 and frac < 0.01, f"hue {fh}: max diff {mx}, {frac:.4%} pixels differ"
    gref = np.asarray(to_grayscale3(Image.fromarray(a)))
    got, _, _ = pipe.run(pipe.upload(a), None, resize=None, plan=[("gray",)])
    assert _diff(got.cpu().numpy(), gref)[0] == 0


@pytest.mark.parametrize("sigma", [0.1, 0.15, 0.2, 1.0])
def test_blur_matches_host(sigma):
    from view_neti_amd.compat.augment import blur_with_sigma
    pipe = _pipe()
    a = _img(97, 131, seed=5)
    ref = np.asarray(blur_with_sigma(Image.fromarray(a), sigma))
    got, _, _ = pipe.run(pipe.upload(a), None, resize=None, plan=[("blur", sigma)])
    torch.cuda.synchronize()
    mx, frac = _diff(got.cpu().numpy(), ref)
    assert mx <= 1 and frac < 1e-3, f"blur {sigma}: max diff {mx}, {frac:.4%}"


@pytest.mark.parametrize("angle", [-10.0, -3.7, 0.0, 0.01, 5.5, 9.99])
@pytest.mark.parametrize("hw", [(384, 512), (512, 512)])
def test_rotation_matches_pillow(angle, hw):
    """Image.rotate(NEAREST, fillcolor) = Geometry.c affine_fixed: bit exact"""
    pipe = _pipe()
    a = _img(*hw, seed=6)
    ref = np.asarray(Image.fromarray(a).rotate(angle, resample=Image.NEAREST, expand=False, fillcolor=(1, 1, 1)))
    got, _, _ = pipe.run(pipe.upload(a), None, resize=None, plan=[("rotate", angle)])
    torch.cuda.synchronize()
    mx, frac = _diff(got.cpu().numpy(), ref)
    assert mx == 0, f"rotate {angle}: max diff {mx}, {frac:.4%}"


@pytest.mark.parametrize("box", [(10, 20, 300, 400), (0, 0, 384, 512), (3, 5, 381, 505), (50, 60, 200, 333)])
def test_random_resized_crop_matches_pillow(box):
    pipe = _pipe()
    a = _img(384, 512, seed=7)
    i, j, h, w = box
    ref = np.asarray(Image.fromarray(a).crop((j, i, j + w, i + h)).resize((512, 384), resample=Image.BILINEAR))
    got, oh, ow = pipe.run(pipe.upload(a), None, resize=None, plan=[("rrcrop", i, j, h, w, 384, 512)])
    torch.cuda.synchronize()
    assert (oh, ow) == (384, 512)
    mx, frac = _diff(got.cpu().numpy(), ref)
    assert mx == 0, f"rrcrop {box}: max diff {mx}, {frac:.4%}"


@pytest.mark.parametrize("key", [1, 2, 3, 4, 5, 6, 7, 8])
def test_whole_pipelines_match_host(key):
    """the eight augmentation pipelines (dataset.py:238-316) end to end on seeded draws: device vs compat/augment.py"""
    from view_neti_amd.compat.augment import apply_plan, draw_plan
    pipe = _pipe()
    src = _img(600, 800, seed=8)
    size = (384, 512)
    base = Image.fromarray(src).resize((size[1], size[0]), resample=Image.BICUBIC)
    up = pipe.upload(src)
    worst, nonempty = 0.0, 0
    for seed in range(12):
        torch.manual_seed(100 * key + seed)
        plan = draw_plan(key, size, size[1], size[0])
        nonempty += bool(plan)
        ref = np.asarray(apply_plan(base, plan)).astype(np.uint8)
        out = torch.zeros(3, *size, device="cuda")
        got, h, w = pipe.run(up, out, resize=size, plan=plan)
        torch.cuda.synchronize()
        assert (h, w) == ref.shape[:2]
        mx, frac = _diff(got.cpu().numpy(), ref)
        # every stage but the hue conversion is exact; a 1-LSB hue difference can be moved / blended by later stages
        assert mx <= 2 and frac < 0.02, f"key {key} seed {seed} plan {[p[0] for p in plan]}: max {mx}, {frac:.3%}"
        worst = max(worst, frac)
        refn = (ref / 127.5 - 1.0).astype(np.float32)
        assert (out.cpu() - torch.from_numpy(refn).permute(2, 0, 1)).abs().max().item() <= 2 / 127.5 + 1e-6
    assert nonempty >= 6
    print(f"key {key}: worst differing-pixel fraction {worst:.4%}")
