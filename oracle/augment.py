"""CPU restatement (numpy, integer / byte arithmetic) of the two PIL-based training augmentations of the reference's image pipeline.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference call sites: `utils/utils.py:412-417` (`BlurPIL(probability=.05, kernel_limits=(3, 7))`,
`ColorJitter(brightness=(2/3, 1.5), contrast=(2/3, 1.5), saturation=(2/3, 1.5), hue=(-.05, .05))`), `utils/transforms.py:242-251`
(`BlurPIL.__call__`: `img.filter(ImageFilter.GaussianBlur(radius=np.random.randint(*kernel_limits)))`).

The arithmetic itself lives in third-party code that is not under /root/reference:
  * Pillow (version unpinned by the reference: `environment.yml` lists bare `pillow`): `ImageFilter.GaussianBlur` = three
    passes of an extended box blur per direction (libImaging/BoxBlur.c: 8.24 fixed point, window weights `ww`, fractional edge
    weights `fw`, edge pixels replicated), `Image.blend` (libImaging/Blend.c: float interpolation truncated to uint8, clipped when
    extrapolating), `convert('L')` (ITU-R 601-2 luma in 16.16 fixed point), `convert('HSV')` / back (libImaging/Convert.c, after
    colorsys.py, mixed float / double arithmetic).
  * torchvision `ColorJitter` on PIL images (version unpinned, NOT installed here): brightness / contrast / saturation =
    `ImageEnhance.{Brightness, Contrast, Color}(img).enhance(f)` = `Image.blend(degenerate, img, f)` with degenerate = black / the
    rounded mean luma / the luma image; hue = add `uint8(f * 255)` to the H channel of the HSV image with uint8 wrap-around.

Pinning: every function below is pinned against Pillow 12.2 in the build container -- RGB->HSV, HSV->RGB and luma EXHAUSTIVELY
over all 2^24 triples, blend over all 2^16 byte pairs for ten factors, the blur for integer radii 1 ... 10 (the reference draws 3 ... 6) on random images down to 3 x 3 pixels
(`tests/golden/make_golden_augment.py` regenerates the committed fixtures `tests/golden/augment.npz` from PIL itself).
PARITY UNPINNED for the ORDER in which torchvision draws the jitter parameters (`ColorJitter.get_params`: a random permutation
of the four operations, then one uniform factor each): `sample_color_jitter` follows the published torchvision >= 0.9 order.
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------------ Gaussian blur (BoxBlur.c)
def gaussian_box_radius(radius, passes=3):
    """ImagingGaussianBlur: the (fractional) box radius whose `passes`-fold convolution has the variance of the Gaussian"""
    sigma2 = float(radius * radius) / passes
    L = math.sqrt(12.0 * sigma2 + 1.0)
    l = math.floor((L - 1.0) / 2.0)
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2)
    a /= 6 * (sigma2 - (l + 1) * (l + 1))
    return float(np.float32(l + a))


def box_weights(fradius):
    radius = int(fradius)
    # (UINT32)(1 << 24) / (floatRadius * 2 + 1): FLOAT arithmetic (double arithmetic is off by one for GaussianBlur(1))
    ww = int(np.float32(1 << 24) / (np.float32(fradius) * np.float32(2) + np.float32(1))) & 0xFFFFFFFF
    fw = (((1 << 24) - (radius * 2 + 1) * ww) // 2) & 0xFFFFFFFF
    return radius, ww, fw


def box_blur_axis(img, fradius, axis):
    """one ImagingLineBoxBlur8 pass along `axis` of a uint8 array: out[x] = (acc(x) * ww + (far_left + far_right) * fw + 2^23) >> 24
    in uint32 arithmetic, acc = sum of the 2 radius + 1 window pixels, far = the two pixels just outside it, indices clamped to
    the line (edge replication) -- the closed form of the running-sum loops of BoxBlur.c"""
    radius, ww, fw = box_weights(fradius)
    a = np.moveaxis(img, axis, -1).astype(np.int64)
    n = a.shape[-1]
    idx = np.arange(n)
    acc = np.zeros_like(a)
    for j in range(-radius, radius + 1):
        acc += a[..., np.clip(idx + j, 0, n - 1)]
    far = a[..., np.clip(idx - radius - 1, 0, n - 1)] + a[..., np.clip(idx + radius + 1, 0, n - 1)]
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    out = (((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24).astype(np.uint8)
    return np.moveaxis(out, -1, axis)


def gaussian_blur(img, radius):
    """`Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius))` for a uint8 [H, W, C] image"""
    fr = gaussian_box_radius(radius)
    out = img
    for _ in range(3):
        out = box_blur_axis(out, fr, 1)
    for _ in range(3):
        out = box_blur_axis(out, fr, 0)
    return out


# ------------------------------------------------------------------------------------------------ blend / luma / HSV
def blend(in1, in2, alpha):
    """Image.blend(im1, im2, alpha) on uint8 arrays (Blend.c)"""
    al = np.float32(alpha)
    t = in1.astype(np.float32) + al * (in2.astype(np.int32) - in1.astype(np.int32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def luma(rgb):
    """convert('L'): ITU-R 601-2, 16.16 fixed point, rounded"""
    r = rgb.astype(np.int64)
    return ((r[..., 0] * 19595 + r[..., 1] * 38470 + r[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def rgb_to_hsv(rgb):
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    with np.errstate(all="ignore"):
        s = cr / maxc.astype(np.float32)
        rc, gc, bc = ((maxc - c).astype(np.float32) / cr for c in (r, g, b))
        h = np.where(r == maxc, (bc - gc).astype(np.float64),
                     np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64), 4.0 + gc.astype(np.float64) - rc.astype(np.float64)))
        h = h.astype(np.float32)
        h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
        uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
        us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    same = minc == maxc
    return np.stack([np.where(same, 0, uh), np.where(same, 0, us), maxc], -1).astype(np.uint8)


def _c_round(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))


def hsv_to_rgb(hsv):
    h = hsv[..., 0].astype(np.float32)
    s = hsv[..., 1]
    vv = hsv[..., 2]
    hd = h.astype(np.float64) * 6.0 / 255.0
    i = np.floor(hd).astype(np.int64)
    f = (hd - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)
    vd = vv.astype(np.float64)
    p = np.clip(_c_round(vd * (1.0 - fs)), 0, 255).astype(np.uint8)
    q = np.clip(_c_round(vd * (1.0 - fs * f)), 0, 255).astype(np.uint8)
    t = np.clip(_c_round(vd * (1.0 - fs * (1.0 - f))), 0, 255).astype(np.uint8)
    k = i % 6
    r = np.choose(k, [vv, q, p, p, t, vv])
    g = np.choose(k, [t, vv, vv, q, p, p])
    b = np.choose(k, [p, p, t, vv, vv, q])
    gray = s == 0
    return np.stack([np.where(gray, vv, r), np.where(gray, vv, g), np.where(gray, vv, b)], -1).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ torchvision ColorJitter on PIL images
BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def adjust(img, op, factor):
    """one operation of torchvision.transforms.functional on a uint8 RGB image [H, W, 3]"""
    if op == BRIGHTNESS:        # ImageEnhance.Brightness: degenerate = black
        return blend(np.zeros_like(img), img, factor)
    if op == CONTRAST:          # ImageEnhance.Contrast: degenerate = int(mean luma + 0.5) everywhere
        L = luma(img)
        mean = int(float(L.astype(np.int64).sum()) / L.size + 0.5)
        return blend(np.full_like(img, mean), img, factor)
    if op == SATURATION:        # ImageEnhance.Color: degenerate = the luma image
        return blend(np.repeat(luma(img)[..., None], 3, -1), img, factor)
    hsv = rgb_to_hsv(img)       # adjust_hue: H += uint8(factor * 255) with wrap-around
    hsv[..., 0] = (hsv[..., 0].astype(np.int64) + int(np.uint8(int(factor * 255) & 0xFF))) & 0xFF
    return hsv_to_rgb(hsv)


def color_jitter(img, order, factors):
    """order: the permutation of (BRIGHTNESS, CONTRAST, SATURATION, HUE) to apply; factors[op]: its factor"""
    for op in order:
        img = adjust(img, int(op), float(factors[int(op)]))
    return img
