"""Fixtures for the PIL-based augmentations (GaussianBlur, ColorJitter's four operations) generated with Pillow itself.
Run in the build container:  python tests/golden/make_golden_augment.py   -> tests/golden/augment.npz
(reference call sites: utils/utils.py:412-417, utils/transforms.py:242-251; torchvision is not installed: the four colour
operations are spelled out with the PIL calls torchvision.transforms.functional makes for PIL images)"""
import os

import numpy as np
from PIL import Image, ImageEnhance, ImageFilter, ImageStat
import PIL

HERE = os.path.dirname(os.path.abspath(__file__))


def tv_adjust(img, op, f):
    if op == 0:
        return ImageEnhance.Brightness(img).enhance(f)
    if op == 1:
        return ImageEnhance.Contrast(img).enhance(f)
    if op == 2:
        return ImageEnhance.Color(img).enhance(f)
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    np_h = ((np_h.astype(np.int64) + (int(f * 255) & 0xFF)) & 0xFF).astype(np.uint8)      # np_h += np.uint8(f * 255), uint8 wrap-around
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def main():
    rng = np.random.default_rng(7)
    out = {"pil_version": np.array(PIL.__version__)}
    # structured + noisy images (saturated colours, greys, extremes) so that every branch of the HSV code is hit
    imgs = rng.integers(0, 256, (3, 29, 37, 3), dtype=np.uint8)
    imgs[1, :10] = rng.integers(0, 256, (10, 37, 1), dtype=np.uint8)          # grey rows (s == 0)
    imgs[1, 10:14] = 255
    imgs[1, 14:18] = 0
    imgs[2] = (np.linspace(0, 255, 29 * 37 * 3).reshape(29, 37, 3)).astype(np.uint8)
    out["imgs"] = imgs
    for r in (3, 4, 5, 6):
        out["blur_r%d" % r] = np.stack([np.asarray(Image.fromarray(im).filter(ImageFilter.GaussianBlur(radius=r))) for im in imgs])
    tiny = rng.integers(0, 256, (2, 4, 5, 3), dtype=np.uint8)                 # lines shorter than the window
    out["tiny"] = tiny
    out["tiny_blur_r6"] = np.stack([np.asarray(Image.fromarray(im).filter(ImageFilter.GaussianBlur(radius=6))) for im in tiny])
    factors = {0: [2 / 3, 1.0, 1.37, 1.5], 1: [2 / 3, 0.9, 1.21, 1.5], 2: [2 / 3, 1.0, 1.4, 1.5], 3: [-0.05, -0.013, 0.0, 0.031, 0.05]}
    for op, fs in factors.items():
        for i, f in enumerate(fs):
            out["op%d_f%d" % (op, i)] = np.array(f)
            out["op%d_out%d" % (op, i)] = np.stack([np.asarray(tv_adjust(Image.fromarray(im), op, f)) for im in imgs])
    # whole jitter sequences: a permutation of the four operations and one factor each, per image
    orders = np.array([[0, 1, 2, 3], [3, 1, 0, 2], [2, 3, 1, 0]])
    facs = np.array([[1.31, 0.77, 1.45, 0.04], [0.7, 1.5, 2 / 3, -0.05], [1.0, 1.1, 0.9, 0.021]])
    seq = []
    for im, order, fc in zip(imgs, orders, facs):
        p = Image.fromarray(im)
        for op in order:
            p = tv_adjust(p, int(op), float(fc[op]))
        seq.append(np.asarray(p))
    out["seq_orders"], out["seq_factors"], out["seq_out"] = orders, facs, np.stack(seq)
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("wrote augment.npz with", len(out), "arrays, Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
