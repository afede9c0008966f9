"""Random clip augmentation on the GPU (the training loader's transform stack, reference dataset/frame.py:78-140,296-323).

The reference builds a torchvision-v2 pipeline and runs it in loader worker processes on fp32 frames, one set of random
parameters per clip, with the event locations travelling along as degenerate bounding boxes.  Here the host only DRAWS the
parameters (a few dozen floats per clip) and moves the labels; the pixels are resampled and colour-jittered by
`stes_augment_u8` on decoded uint8 frames that are already in HBM.

Geometry: every option of the two RandomChoice lists (frame.py:83-105) and the closing Resize (:107) is a plane homography, so
a clip's geometry is the product of its stages and the frames are resampled ONCE (bilinear) instead of up to three times.
Labels go through the stages one by one and are clamped to the canvas after each, as torchvision does with boxes.
Parameter distributions follow torchvision's `get_params` / `make_params` of the named transforms (the package is not
installed here; results are not compared with it).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # reference dataset/frame.py:16-17
IMAGENET_STD = (0.229, 0.224, 0.225)
N_PARAMS = 32


def _translate(tx, ty):
    return np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float64)


def _scale(sx, sy):
    return np.array([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], np.float64)


def _about_centre(m, w, h):
    return _translate(w / 2, h / 2) @ m @ _translate(-w / 2, -h / 2)


def _rotation(deg):
    """Rotation by `deg` in image coordinates (y down): the matrix torchvision's _get_inverse_affine_matrix inverts."""
    r = math.radians(deg)
    return np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]], np.float64)


def _homography(src, dst):
    """3x3 map with dst_i ~ H src_i for four point pairs (torchvision's _get_perspective_coeffs solves the same system)."""
    a, b = [], []
    for (x, y), (u, v) in zip(src, dst):
        a.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        a.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    c = np.linalg.lstsq(np.asarray(a, np.float64), np.asarray(b, np.float64), rcond=None)[0]
    return np.append(c, 1.0).reshape(3, 3)


class ClipAugmenter:
    """`aug(frames_u8, xy)` -> (normalised fp32 frames [B,T,3,crop,crop], transformed xy), like one pass of the reference's
    `_get_img_transforms(crop_dim, is_eval)` over every clip of the batch (is_eval: Resize + Normalize only)."""

    GEO1 = ("crop", "resized_crop", "resize")                                           # frame.py:84-90
    GEO2 = ("hflip", "zoom_out", "perspective", "rotation", "affine", "none")           # frame.py:93-104

    def __init__(self, crop_dim=224, is_eval=False, seed=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        self.crop_dim, self.is_eval = int(crop_dim), bool(is_eval)
        self.p_color = 0.0 if is_eval else 0.25                                         # frame.py:79
        self.rng = np.random.default_rng(seed)
        self.mean, self.std = tuple(mean), tuple(std)
        self._ws = None

    # ------------------------------------------------------------------ parameter draws
    def _geo1(self, kind, w, h):
        d, r = self.crop_dim, self.rng
        if kind == "crop":                                  # RandomCrop.make_params
            if w < d or h < d:
                raise ValueError(f"RandomCrop: crop size {(d, d)} is larger than the frame {(h, w)}")
            left, top = int(r.integers(0, w - d + 1)), int(r.integers(0, h - d + 1))
            return _translate(-left, -top)
        if kind == "resized_crop":                          # RandomResizedCrop.make_params, scale (0.08, 1), ratio (3/4, 4/3)
            area = h * w
            lo, hi = math.log(3 / 4), math.log(4 / 3)
            for _ in range(10):
                target = area * r.uniform(0.08, 1.0)
                aspect = math.exp(r.uniform(lo, hi))
                cw, ch = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
                if 0 < cw <= w and 0 < ch <= h:
                    top, left = int(r.integers(0, h - ch + 1)), int(r.integers(0, w - cw + 1))
                    break
            else:                                           # central crop at the nearest allowed ratio
                ratio = w / h
                if ratio < 3 / 4:
                    cw, ch = w, int(round(w / (3 / 4)))
                elif ratio > 4 / 3:
                    ch, cw = h, int(round(h * (4 / 3)))
                else:
                    cw, ch = w, h
                top, left = (h - ch) // 2, (w - cw) // 2
            return _scale(d / cw, d / ch) @ _translate(-left, -top)
        return _scale(d / w, d / h)                         # Resize: squash the whole frame

    def _geo2(self, kind, w, h):
        """-> (map, canvas width, canvas height) of the second stage on a w x h canvas."""
        r = self.rng
        if kind == "hflip":
            return np.array([[-1, 0, w], [0, 1, 0], [0, 0, 1]], np.float64), w, h
        if kind == "zoom_out":                              # RandomZoomOut.make_params, side_range (1, 4), fill 0
            k = r.uniform(1.0, 4.0)
            cw, ch = int(w * k), int(h * k)
            u = r.random(2)
            return _translate(int((cw - w) * u[0]), int((ch - h) * u[1])), cw, ch
        if kind == "perspective":                           # RandomPerspective.make_params, distortion_scale 0.6
            bw, bh = int(0.6 * (w // 2)) + 1, int(0.6 * (h // 2)) + 1
            ri = lambda lo, hi: int(r.integers(lo, hi))
            start = [(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)]
            end = [(ri(0, bw), ri(0, bh)), (ri(w - bw, w), ri(0, bh)), (ri(w - bw, w), ri(h - bh, h)), (ri(0, bw), ri(h - bh, h))]
            return _homography(start, end), w, h
        if kind == "rotation":                              # RandomRotation(degrees=(0, 45)): counter-clockwise on screen
            return _about_centre(_rotation(-r.uniform(0.0, 45.0)), w, h), w, h
        if kind == "affine":                                # RandomAffine(degrees=(30,70), translate=(0.1,0.3), scale=(0.5,0.75))
            ang = r.uniform(30.0, 70.0)
            tx = int(round(r.uniform(-0.1 * w, 0.1 * w)))
            ty = int(round(r.uniform(-0.3 * h, 0.3 * h)))
            s = r.uniform(0.5, 0.75)
            return _translate(tx, ty) @ _about_centre(_rotation(ang) @ _scale(s, s), w, h), w, h
        return np.eye(3), w, h

    def sample(self, n_clips, src_hw, geo1=None, geo2=None):
        """Draw the parameters of n_clips clips of h x w frames.  -> (params [n,32] float32, stages): stages[i] is the list of
        (map, canvas_w, canvas_h) the labels of clip i walk through.  geo1 / geo2 force a choice (tests)."""
        h, w = src_hw
        d = self.crop_dim
        params = np.zeros((n_clips, N_PARAMS), np.float32)
        stages = []
        for i in range(n_clips):
            st = []
            if not self.is_eval:
                k1 = geo1 or self.GEO1[int(self.rng.integers(0, 3))]
                st.append((self._geo1(k1, w, h), d, d))
                k2 = geo2 or self.GEO2[int(self.rng.integers(0, 6))]
                st.append(self._geo2(k2, d, d))
                cw, ch = st[-1][1], st[-1][2]
            else:
                cw, ch = w, h
            st.append((_scale(d / cw, d / ch), d, d))       # the closing Resize (frame.py:107)
            fwd = np.eye(3)
            for m, _, _ in st:
                fwd = m @ fwd
            inv = np.linalg.inv(fwd)
            params[i, 0:9] = (inv / inv[2, 2]).reshape(9)
            on = lambda: self.rng.random() < self.p_color   # RandomApply(p) (frame.py:111-123)
            if on():
                params[i, 10:12] = 1.0, self.rng.uniform(-0.2, 0.2)
            if on():
                params[i, 12:14] = 1.0, self.rng.uniform(0.7, 1.2)
            if on():
                params[i, 14:16] = 1.0, self.rng.uniform(0.7, 1.2)
            if on():
                params[i, 16:18] = 1.0, self.rng.uniform(0.7, 1.2)
            if on():
                params[i, 18] = 1.0
                params[i, 19:24] = gaussian_taps(self.rng.uniform(0.1, 2.0))
            params[i, 24:27], params[i, 27:30] = self.mean, self.std
            stages.append(st)
        return params, stages

    # ------------------------------------------------------------------ labels
    def transform_xy(self, xy, stages, src_hw):
        """xy [T,2] as fractions of (w, h) (label files, frame.py:285) -> fractions of crop_dim after the clip's stages
        (frame.py:300-318: points as XYXY boxes, transformed with the frames, clamped to the canvas after every transform)."""
        h, w = src_hw
        pts = np.asarray(xy, np.float64) * np.array([w, h], np.float64)
        for m, cw, ch in stages:
            q = np.concatenate([pts, np.ones((len(pts), 1))], 1) @ m.T
            pts = q[:, :2] / q[:, 2:3]
            pts[:, 0] = np.clip(pts[:, 0], 0, cw)
            pts[:, 1] = np.clip(pts[:, 1], 0, ch)
        return (pts / self.crop_dim).astype(np.float32)

    # ------------------------------------------------------------------ pixels
    def apply(self, frames_u8: torch.Tensor, params, u8=False):
        """frames_u8 [B,T,3,H,W] uint8 on the GPU, params [B,32] -> [B,T,3,crop,crop] (fp32 normalised, or uint8)."""
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 5 or frames_u8.shape[2] != 3 or not frames_u8.is_cuda:
            raise lib.StesError("augmentation takes decoded uint8 clips [B,T,3,H,W] on the GPU")
        B, T, _, H, W = frames_u8.shape
        d = self.crop_dim
        dev = frames_u8.device
        x = frames_u8.contiguous()
        par = torch.as_tensor(np.ascontiguousarray(params, np.float32)).to(dev, non_blocking=True)
        if par.shape != (B, N_PARAMS):
            raise lib.StesError(f"params must be [{B}, {N_PARAMS}]")
        need = lib.lib().stes_augment_ws_floats(B * T, d, d)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.float32, device=dev)
        out = torch.empty((B, T, 3, d, d), dtype=torch.uint8 if u8 else torch.float32, device=dev)
        lib.call("stes_augment_u8", lib.ptr(x), None if u8 else lib.ptr(out), lib.ptr(out) if u8 else None, lib.ptr(par),
                 lib.ptr(self._ws), self._ws.numel(), B, T, H, W, d, d)
        return out

    def __call__(self, frames_u8, xy=None, u8=False):
        B, T, _, H, W = frames_u8.shape
        params, stages = self.sample(B, (H, W))
        out = self.apply(frames_u8, params, u8=u8)
        if xy is None:
            return out, None
        xy_np = xy.detach().cpu().numpy() if torch.is_tensor(xy) else np.asarray(xy)
        new = np.stack([self.transform_xy(xy_np[b], stages[b], (H, W)) for b in range(B)])
        return out, (torch.as_tensor(new).to(xy.device) if torch.is_tensor(xy) else new)


def gaussian_taps(sigma):
    """Five normalised taps of torchvision's GaussianBlur(5) for one sigma (_get_gaussian_kernel1d)."""
    x = np.linspace(-2.0, 2.0, 5, dtype=np.float32)
    pdf = np.exp(-0.5 * (x / np.float32(sigma)) ** 2).astype(np.float32)
    return (pdf / pdf.sum()).astype(np.float32)
