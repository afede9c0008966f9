"""CPU tests of the augmentation oracle and of the host side of spot_amd.augment (parameter draws, label transport).

torchvision is not installed, so the oracle is pinned against the torch primitives torchvision's kernels are built on
(`grid_sample` with align_corners=False and zero padding, `conv2d` over a reflect-padded image) and against closed-form cases.
"""
import numpy as np
import torch
import torch.nn.functional as TF

from oracle import augment_oracle as ao
from spot_amd.augment import ClipAugmenter, gaussian_taps


def _frame(h=40, w=56, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (3, h, w), dtype=np.uint8)


def test_warp_matches_grid_sample():
    """A random affine map: the oracle's resampling against torch's grid_sample fed with the same source coordinates."""
    f = _frame()
    _, hs, ws = f.shape
    ho, wo = 32, 48
    hinv = np.array([[0.9, 0.2, 3.0], [-0.15, 1.1, 2.0], [0, 0, 1]], np.float32)
    got = ao.warp(f, hinv, ho, wo)
    oy, ox = np.meshgrid(np.arange(ho), np.arange(wo), indexing="ij")
    px, py = ox + 0.5, oy + 0.5
    sx = hinv[0, 0] * px + hinv[0, 1] * py + hinv[0, 2]
    sy = hinv[1, 0] * px + hinv[1, 1] * py + hinv[1, 2]
    grid = torch.tensor(np.stack([sx / (0.5 * ws) - 1, sy / (0.5 * hs) - 1], -1), dtype=torch.float32)[None]
    ref = TF.grid_sample(torch.tensor(f, dtype=torch.float32)[None] / 255, grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0]
    assert np.abs(got - ref.numpy()).max() < 2e-5


def test_identity_crop_flip_are_exact():
    f = _frame(32, 48)
    v = f.astype(np.float32) / np.float32(255)
    assert np.array_equal(ao.warp(f, np.eye(3), 32, 48), v)
    crop = np.array([[1, 0, 5], [0, 1, 3], [0, 0, 1]], np.float32)           # output (x, y) reads source (x + 5, y + 3)
    assert np.array_equal(ao.warp(f, crop, 16, 16), v[:, 3:19, 5:21])
    flip = np.array([[-1, 0, 48], [0, 1, 0], [0, 0, 1]], np.float32)
    assert np.array_equal(ao.warp(f, flip, 32, 48), v[:, :, ::-1])
    assert np.array_equal(ao.warp(f, flip, 32, 48, nearest=True), v[:, :, ::-1])


def test_colour_steps():
    img = (_frame(16, 16, 3).astype(np.float32) / np.float32(255))
    # a zero hue shift goes through HSV and back
    assert np.abs(ao.hue_shift(img, 0.0) - img).max() < 1e-5
    # a shift by a third of the circle turns red into green
    red = np.zeros((3, 1, 1), np.float32); red[0] = 0.8
    out = ao.hue_shift(red, 1.0 / 3.0)
    assert abs(out[1, 0, 0] - 0.8) < 1e-5 and out[0, 0, 0] < 1e-5 and out[2, 0, 0] < 1e-5
    g = ao.grey_of(img)
    assert np.allclose(ao.blend(img, g[None], 1.0), img) and np.allclose(ao.blend(img, g[None], 0.0), np.broadcast_to(g, img.shape))
    assert np.allclose(ao.blend(img, 0 * img, 1.2), np.clip(1.2 * img, 0, 1), atol=1e-6)


def test_blur_matches_conv2d():
    img = _frame(20, 24, 5)
    p = np.zeros(32, np.float32)
    p[0:9] = np.eye(3).reshape(9)
    p[18] = 1
    p[19:24] = gaussian_taps(1.3)
    p[27:30] = 1
    got = ao.augment_frame(img, p, 20, 24)
    k = torch.tensor(p[19:24])
    k2 = (k[:, None] * k[None, :])[None, None].repeat(3, 1, 1, 1)
    x = TF.pad(torch.tensor(img, dtype=torch.float32)[None] / 255, (2, 2, 2, 2), mode="reflect")
    ref = TF.conv2d(x, k2, groups=3)[0].numpy()
    assert np.abs(got - ref).max() < 1e-5
    assert abs(float(gaussian_taps(0.7).sum()) - 1) < 1e-6 and np.array_equal(gaussian_taps(0.7), ao.gaussian_taps(0.7))


def test_params_invert_the_stages_and_labels_follow():
    aug = ClipAugmenter(64, seed=3)
    h, w = 72, 128
    for g2 in ClipAugmenter.GEO2:
        for g1 in ClipAugmenter.GEO1:
            params, stages = aug.sample(1, (h, w), geo1=g1, geo2=g2)
            fwd = np.eye(3)
            for m, _, _ in stages[0]:
                fwd = m @ fwd
            inv = params[0, :9].reshape(3, 3).astype(np.float64)
            prod = inv @ fwd
            assert np.allclose(prod / prod[2, 2], np.eye(3), atol=2e-4), (g1, g2)
            assert stages[0][-1][1:] == (64, 64)
            # a label inside the frame ends inside the canvas; the unclamped map agrees with the stage walk when nothing clamps
            xy = np.array([[0.5, 0.5]], np.float32)
            out = aug.transform_xy(xy, stages[0], (h, w))
            assert (out >= 0).all() and (out <= 1).all()
    # closed forms: crop then flip
    aug = ClipAugmenter(64, seed=0)
    params, stages = aug.sample(1, (h, w), geo1="crop", geo2="hflip")
    left, top = -stages[0][0][0][0, 2], -stages[0][0][0][1, 2]
    xy = np.array([[(left + 10) / w, (top + 20) / h]], np.float32)
    assert np.allclose(aug.transform_xy(xy, stages[0], (h, w)), [[(64 - 10) / 64, 20 / 64]], atol=1e-6)
    # a point left of the crop is clamped to the border before the flip, like a clamped box
    xy = np.array([[0.0, (top + 20) / h]], np.float32)
    if left > 0:
        assert np.allclose(aug.transform_xy(xy, stages[0], (h, w)), [[1.0, 20 / 64]], atol=1e-6)


def test_eval_mode_is_resize_and_normalize_only():
    aug = ClipAugmenter(32, is_eval=True, seed=1)
    params, stages = aug.sample(3, (48, 64))
    assert len(stages[0]) == 1 and not params[:, [9, 10, 12, 14, 16, 18]].any()
    assert np.allclose(params[0, :9].reshape(3, 3), [[2, 0, 0], [0, 1.5, 0], [0, 0, 1]])
    assert np.allclose(params[0, 24:27], (0.485, 0.456, 0.406)) and np.allclose(params[0, 27:30], (0.229, 0.224, 0.225))


def test_choice_and_jitter_rates():
    """RandomChoice is uniform over its lists and each colour step fires with p = 0.25 (dataset/frame.py:79,83-123)."""
    aug = ClipAugmenter(32, seed=7)
    params, stages = aug.sample(4000, (40, 40))
    for col in (10, 12, 14, 16, 18):
        assert abs(params[:, col].mean() - 0.25) < 0.03
    assert (np.abs(params[params[:, 10] != 0, 11]) <= 0.2).all()
    for col in (13, 15, 17):
        on = params[params[:, col - 1] != 0, col]
        assert (on >= 0.7).all() and (on <= 1.2).all()
    zoomed = np.mean([st[1][1] > 32 for st in stages])           # zoom-out is the only stage that grows the canvas
    assert abs(zoomed - 1 / 6) < 0.03
