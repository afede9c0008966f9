"""GPU tests of stes_augment_u8 (through the C-ABI, spot_amd.augment.ClipAugmenter.apply) against the numpy oracle.

torchvision is absent, so parity with the reference's torchvision pipeline is unpinned; the oracle restates torchvision's
published kernels (see oracle/augment_oracle.py) and is itself pinned against torch's grid_sample / conv2d in
tests/test_augment_host.py.  Tolerance: 2e-5 on values in [0,1] (fp32 arithmetic in the same order on both sides; the frame
mean of the contrast step is summed in a different order), 1 LSB on uint8 output.
"""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as ao
from spot_amd.augment import ClipAugmenter, gaussian_taps

pytestmark = pytest.mark.gpu


def _clips(B, T, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    # smooth-ish content (random low-resolution images upsampled) plus noise, so that interpolation errors would show
    low = torch.rand(B * T, 3, H // 4 + 1, W // 4 + 1, generator=g)
    img = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear") * 0.8 + 0.2 * torch.rand(B * T, 3, H, W, generator=g)
    return (img * 255).round().clamp(0, 255).to(torch.uint8).reshape(B, T, 3, H, W)


def _denorm(out, p):
    return out * p[27:30, None, None] + p[24:27, None, None]


def test_identity_crop_flip_exact():
    B, T, H, W, d = 3, 2, 48, 80, 48
    x = _clips(B, T, H, W)
    aug = ClipAugmenter(d, seed=0)
    params = np.zeros((B, 32), np.float32)
    params[:, 24:27], params[:, 27:30] = aug.mean, aug.std
    params[0, :9] = np.array([[1, 0, 7], [0, 1, 0], [0, 0, 1]], np.float32).reshape(9)          # crop at x = 7
    params[1, :9] = np.array([[-1, 0, 7 + d], [0, 1, 0], [0, 0, 1]], np.float32).reshape(9)     # the same crop, mirrored
    params[2, :9] = np.array([[1, 0, 7], [0, 1, 0], [0, 0, 1]], np.float32).reshape(9)
    params[2, 9] = 1                                                                                # nearest
    out_u8 = aug.apply(x.cuda(), params, u8=True).cpu()
    assert torch.equal(out_u8[0], x[0, :, :, :, 7:7 + d])
    assert torch.equal(out_u8[1], x[1, :, :, :, 7:7 + d].flip(-1))
    assert torch.equal(out_u8[2], x[2, :, :, :, 7:7 + d])
    out = aug.apply(x.cuda(), params).cpu().numpy()
    want = (x[0, :, :, :, 7:7 + d].numpy().astype(np.float32) / np.float32(255) - params[0, 24:27, None, None]) / params[0, 27:30, None, None]
    assert np.array_equal(out[0], want.astype(np.float32))


@pytest.mark.parametrize("geo2", ClipAugmenter.GEO2)
def test_against_oracle_every_geometry(geo2):
    B, T, H, W, d = 3, 2, 72, 128, 64
    x = _clips(B, T, H, W, seed=1)
    aug = ClipAugmenter(d, seed=11)
    params = np.concatenate([aug.sample(1, (H, W), geo1=g1, geo2=geo2)[0] for g1 in ClipAugmenter.GEO1])
    # every colour step on, with drawn strengths
    r = np.random.default_rng(5)
    params[:, 10], params[:, 11] = 1, r.uniform(-0.2, 0.2, B)
    for c in (12, 14, 16):
        params[:, c], params[:, c + 1] = 1, r.uniform(0.7, 1.2, B)
    params[:, 18] = 1
    for b in range(B):
        params[b, 19:24] = gaussian_taps(r.uniform(0.1, 2.0))
    out = aug.apply(x.cuda(), params).cpu().numpy().reshape(B * T, 3, d, d)
    ref = ao.augment(x.numpy().reshape(B * T, 3, H, W), params, T, d, d)
    for i in range(B * T):
        p = params[i // T]
        diff = np.abs(_denorm(out[i], p) - _denorm(ref[i], p))
        # (a pixel on the border between two hue sectors may take the other branch: allow a handful)
        assert (diff > 2e-5).mean() < 1e-3 and np.median(diff) < 1e-6, (geo2, i, diff.max())
    out8 = aug.apply(x.cuda(), params, u8=True).cpu().numpy().reshape(B * T, 3, d, d).astype(np.int32)
    ref8 = ao.augment(x.numpy().reshape(B * T, 3, H, W), params, T, d, d, u8=True).astype(np.int32)
    assert np.abs(out8 - ref8).max() <= 1 and (out8 != ref8).mean() < 2e-3


def test_each_colour_step_alone_and_determinism():
    B, T, H, W, d = 5, 3, 64, 64, 64
    x = _clips(B, T, H, W, seed=2)
    aug = ClipAugmenter(d, seed=0)
    params = np.zeros((B, 32), np.float32)
    params[:, :9] = np.eye(3, dtype=np.float32).reshape(9)
    params[:, 24:27], params[:, 27:30] = 0.0, 1.0
    params[0, 10:12] = 1, 0.13
    params[1, 12:14] = 1, 0.75
    params[2, 14:16] = 1, 1.15
    params[3, 16:18] = 1, 0.8
    params[4, 18], params[4, 19:24] = 1, gaussian_taps(0.9)
    a = aug.apply(x.cuda(), params)
    b = aug.apply(x.cuda(), params)
    assert torch.equal(a, b)
    ref = ao.augment(x.numpy().reshape(B * T, 3, H, W), params, T, d, d)
    diff = np.abs(a.cpu().numpy().reshape(B * T, 3, d, d) - ref)
    assert (diff > 2e-5).mean() < 1e-3 and diff.max() < 0.7, diff.max()
    assert np.median(diff) == 0


@pytest.mark.parametrize("geo2", ["zoom_out", "perspective", "rotation", "affine", "hflip"])
def test_labels_travel_with_the_pixels(geo2):
    """A bright blob at the labelled position of a dark frame lands where the transformed label says."""
    H, W, d = 96, 160, 64
    aug = ClipAugmenter(d, seed=21)
    hits = 0
    for trial in range(6):
        params, stages = aug.sample(1, (H, W), geo1="resize", geo2=geo2)
        params[0, 10:19] = 0
        xy = np.array([[0.35 + 0.05 * trial, 0.4 + 0.04 * trial]], np.float32)
        cx, cy = int(xy[0, 0] * W), int(xy[0, 1] * H)
        x = torch.zeros(1, 1, 3, H, W, dtype=torch.uint8)
        x[..., cy - 4:cy + 5, cx - 4:cx + 5] = 255
        out = aug.apply(x.cuda(), params, u8=True).cpu().numpy()[0, 0, 0].astype(np.float64)
        want = aug.transform_xy(np.array([[(cx + 0.5) / W, (cy + 0.5) / H]], np.float32), stages[0], (H, W))[0] * d
        if out.sum() == 0 or not (2 < want[0] < d - 2 and 2 < want[1] < d - 2):
            continue                              # the blob left the canvas (and the label was clamped to its border)
        ys, xs = np.mgrid[0:d, 0:d]
        got = np.array([(out * (xs + 0.5)).sum(), (out * (ys + 0.5)).sum()]) / out.sum()
        assert np.abs(got - want).max() < 1.0, (geo2, trial, got, want)
        hits += 1
    assert hits >= 2


def test_full_batch_feeds_the_model_input_format():
    """The configuration's batch: 16 clips x 64 frames of 398x224 decoded video -> [16,64,3,224,224] fp32, finite, normalised."""
    B, T, H, W = 16, 64, 224, 398
    x = torch.randint(0, 256, (B, T, 3, H, W), dtype=torch.uint8, device="cuda")
    aug = ClipAugmenter(224, seed=4)
    xy = torch.rand(B, T, 2)
    out, xy2 = aug(x, xy)
    assert out.shape == (B, T, 3, 224, 224) and out.dtype == torch.float32 and bool(torch.isfinite(out).all())
    assert xy2.shape == xy.shape and float(xy2.min()) >= 0 and float(xy2.max()) <= 1
    lo = (0 - np.array(aug.mean)) / np.array(aug.std)
    hi = (1 - np.array(aug.mean)) / np.array(aug.std)
    for c in range(3):
        assert float(out[:, :, c].min()) >= lo[c] - 1e-4 and float(out[:, :, c].max()) <= hi[c] + 1e-4
    # evaluation clips: plain Resize + Normalize
    ev = ClipAugmenter(224, is_eval=True)
    o2, _ = ev(x[:1, :2])
    p, _ = ev.sample(1, (H, W))
    ref = ao.augment(x[:1, :2].cpu().numpy().reshape(2, 3, H, W), p, 2, 224, 224)
    assert np.abs(o2.cpu().numpy().reshape(2, 3, 224, 224) - ref).max() < 1e-4


def test_train_driver_with_augment_on_shards_of_another_size(tmp_path):
    """spot_amd.train --augment: shards hold 80x112 frames, the model trains on augmented 64x64 crops (both the reference-style
    epoch loop and the fused step), validation clips are resized."""
    import os
    from spot_amd import train as TR
    shard_dir = tmp_path / "shards"
    os.makedirs(shard_dir)
    g = torch.Generator().manual_seed(0)
    for split in ("train", "val"):
        np.savez(shard_dir / f"{split}_000.npz", frame=torch.randint(0, 256, (4, 4, 3, 80, 112), generator=g, dtype=torch.uint8).numpy(),
                 label=(torch.rand(4, 4, generator=g) < 0.3).long().numpy() * 2, xy=torch.rand(4, 4, 2, generator=g).numpy())
    base = ["ds", str(shard_dir), "-m", "rny008_gsm", "-t", "gru", "--clip_len", "4", "--crop_dim", "64", "--batch_size", "2",
            "--warm_up_epochs", "1", "--predict_location", "--num_classes", "6", "--criterion", "loss", "--num_epochs", "2", "--augment"]
    _, losses, _ = TR.main(TR.get_args(base + ["-s", str(tmp_path / "a")]))
    assert len(losses) == 2 and all(np.isfinite(l["train"]) and np.isfinite(l["val"]) for l in losses)
    _, losses, _ = TR.main(TR.get_args(base + ["-s", str(tmp_path / "b"), "--fused"]))
    assert len(losses) == 2 and all(np.isfinite(l["train"]) and np.isfinite(l["val"]) for l in losses)
