"""CPU restatement of the training loader's clip augmentation -- TEST INFRASTRUCTURE, not a product path.

Only tests/ may import this module.  It restates, in numpy fp32, what reference dataset/frame.py:78-140 asks torchvision's
transforms.v2 to do to one clip (all frames with the same random parameters, dataset/frame.py:169-170,296-323):

* geometry (frame.py:81-107): every choice in the two RandomChoice lists and the closing Resize is a plane homography, so a
  clip's geometry is one 3x3 map.  `warp` resamples through it the way torchvision's grid-based kernels do
  (functional/_geometry.py `_apply_grid_transform`: grid_sample, bilinear, align_corners=False, zeros outside);
* colour (frame.py:109-125): `adjust_hue` (`_rgb_to_hsv` / `_hsv_to_rgb`), `adjust_saturation`, `adjust_brightness`,
  `adjust_contrast` (all `_blend`s, functional/_color.py), `gaussian_blur` (5 taps per axis, reflect padding,
  functional/_misc.py), `Normalize`.

PARITY UNPINNED against torchvision itself: the package (requirements.txt:2, `torchvision>=0.13`, no lockfile) is not installed
here and the reference holds no augmented fixtures.  What the tests pin instead: the GPU kernels against this restatement, and
size-independent properties (identity, integer crops and flips are exact copies; a marked pixel lands where the transformed
label says).  Known, deliberate differences from running the torchvision pipeline: the geometric steps are composed into ONE
bilinear resampling (torchvision resamples up to three times, with antialiasing in Resize and nearest-neighbour in
RandomRotation / RandomAffine).
"""
import numpy as np

F = np.float32


def _fetch(plane, x, y):
    H, W = plane.shape
    ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
    v = plane[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)].astype(F)      # the byte; /255 follows the interpolation
    return np.where(ok, v, F(0.0)).astype(F)


def warp(frame_u8, hinv, Ho, Wo, nearest=False):
    """frame_u8 [3,Hs,Ws] uint8 -> [3,Ho,Wo] fp32 in [0,1]; hinv maps output (index + 0.5) to source (index + 0.5) coordinates."""
    _, Hs, Ws = frame_u8.shape
    h = np.asarray(hinv, F).reshape(9)
    oy, ox = np.meshgrid(np.arange(Ho), np.arange(Wo), indexing="ij")
    px, py = ox.astype(F) + F(0.5), oy.astype(F) + F(0.5)
    with np.errstate(all="ignore"):
        w = (h[6] * px + h[7] * py).astype(F) + h[8]
        sx = ((h[0] * px + h[1] * py).astype(F) + h[2]) / w - F(0.5)
        sy = ((h[3] * px + h[4] * py).astype(F) + h[5]) / w - F(0.5)
        inside = (sx > -2) & (sx < Ws + 1) & (sy > -2) & (sy < Hs + 1)
    sx = np.where(inside, sx, F(0)).astype(F)
    sy = np.where(inside, sy, F(0)).astype(F)
    out = np.zeros((3, Ho, Wo), F)
    if nearest:
        xi, yi = np.rint(sx).astype(np.int64), np.rint(sy).astype(np.int64)
        for c in range(3):
            out[c] = _fetch(frame_u8[c], xi, yi) / F(255.0)
    else:
        x0f, y0f = np.floor(sx), np.floor(sy)
        x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
        wx, wy = (sx - x0f).astype(F), (sy - y0f).astype(F)
        one = F(1.0)
        w00, w01, w10, w11 = (one - wx) * (one - wy), wx * (one - wy), (one - wx) * wy, wx * wy
        for c in range(3):
            p = frame_u8[c]
            out[c] = (((_fetch(p, x0, y0) * w00 + _fetch(p, x0 + 1, y0) * w01) + _fetch(p, x0, y0 + 1) * w10) + _fetch(p, x0 + 1, y0 + 1) * w11) / F(255.0)
    return np.where(inside[None], out, F(0)).astype(F)


def grey_of(img):
    return ((F(0.2989) * img[0] + F(0.587) * img[1]).astype(F) + F(0.114) * img[2]).astype(F)


def blend(a, b, ratio):
    ratio = F(ratio)
    return np.clip((ratio * a).astype(F) + ((F(1.0) - ratio) * b).astype(F), F(0), F(1)).astype(F)


def hue_shift(img, shift):
    r, g, b = img[0], img[1], img[2]
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    eq = maxc == minc
    cr = (maxc - minc).astype(F)
    s = (cr / np.where(eq, F(1), maxc)).astype(F)
    div = np.where(eq, F(1), cr).astype(F)
    rc, gc, bc = ((maxc - r) / div).astype(F), ((maxc - g) / div).astype(F), ((maxc - b) / div).astype(F)
    hr = np.where(maxc == r, bc - gc, F(0)).astype(F)
    hg = np.where((maxc == g) & (maxc != r), (F(2) + rc).astype(F) - bc, F(0)).astype(F)
    hb = np.where((maxc != g) & (maxc != r), (F(4) + gc).astype(F) - rc, F(0)).astype(F)
    h = np.fmod((((hr + hg).astype(F) + hb).astype(F) / F(6)).astype(F) + F(1), F(1)).astype(F)
    h = (h + F(shift)).astype(F)
    h = (h - np.floor(h)).astype(F)
    h6 = (h * F(6)).astype(F)
    fl = np.floor(h6)
    f = (h6 - fl).astype(F)
    i = fl.astype(np.int64) % 6
    v = maxc
    one = F(1)
    p = np.clip(v * (one - s), 0, 1).astype(F)
    q = np.clip(v * (one - (s * f).astype(F)), 0, 1).astype(F)
    t = np.clip(v * (one - (s * (one - f)).astype(F)), 0, 1).astype(F)
    sel = lambda *opts: np.select([i == k for k in range(6)], opts).astype(F)
    return np.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)])


def gaussian_taps(sigma):
    """torchvision _get_gaussian_kernel1d for kernel size 5."""
    x = np.linspace(-2.0, 2.0, 5, dtype=np.float32)
    pdf = np.exp(-0.5 * (x / np.float32(sigma)) ** 2).astype(F)
    return (pdf / pdf.sum()).astype(F)


def augment_frame(frame_u8, p, Ho, Wo):
    """One frame through the whole chain; p is the 32-float parameter row of its clip.  Returns the value in [0,1] before Normalize."""
    img = warp(frame_u8, p[0:9], Ho, Wo, nearest=p[9] != 0)
    if p[10] != 0:
        img = hue_shift(img, p[11])
    if p[12] != 0:
        img = blend(img, grey_of(img)[None], p[13])
    if p[14] != 0:
        img = blend(img, np.zeros_like(img), p[15])
    if p[16] != 0:
        # (the kernel adds up 256-pixel partial sums in order; fp64 here, the difference is below the test tolerance)
        mean = F(grey_of(img).astype(np.float64).sum() / (Ho * Wo))
        img = blend(img, mean, p[17])
    if p[18] != 0:
        k = np.asarray(p[19:24], F)
        pad = np.pad(img, ((0, 0), (2, 2), (2, 2)), mode="reflect")
        acc = np.zeros_like(img)
        for dy in range(5):
            for dx in range(5):
                acc = (acc + (k[dy] * k[dx]).astype(F) * pad[:, dy:dy + Ho, dx:dx + Wo]).astype(F)
        img = acc
    return img


def augment(frames_u8, params, T, Ho, Wo, u8=False):
    """frames_u8 [n,3,Hs,Ws], params [n_clips,32] -> [n,3,Ho,Wo] normalised fp32, or uint8 when u8."""
    n = frames_u8.shape[0]
    out = np.zeros((n, 3, Ho, Wo), np.uint8 if u8 else F)
    for i in range(n):
        p = np.asarray(params[i // T], F)
        v = augment_frame(frames_u8[i], p, Ho, Wo)
        if u8:
            out[i] = np.rint(np.clip(v, 0, 1) * F(255)).astype(np.uint8)
        else:
            out[i] = ((v - p[24:27, None, None]) / p[27:30, None, None]).astype(F)
    return out
