"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch) of ttach.SegmentationTTAWrapper as BaseManager.infer builds it
(managers/BaseManager.py:652-660): HorizontalFlip x Scale([0.75, 1, 1.5, 1.75, 2]), merge_mode='mean'.
ttach is not vendored / installed: parity unpinned, restated from its published source (ttach 0.0.3 functional.scale,
transforms.HorizontalFlip / Scale, base.Compose / Merger)."""
import itertools

import torch
import torch.nn.functional as F


def _scale(x, s):
    h, w = x.shape[2:]
    return F.interpolate(x, size=(int(h * s), int(w * s)), mode="nearest")


def tta_forward(model_fn, image, scales=(0.75, 1, 1.5, 1.75, 2), flips=(False, True)):
    total, n = None, 0
    for flip, s in itertools.product(flips, scales):
        x = image.flip(3) if flip else image
        if s != 1:
            x = _scale(x, s)
        y = model_fn(x)
        if s != 1:
            y = _scale(y, 1 / s)
        if flip:
            y = y.flip(3)
        total = y if total is None else total + y
        n += 1
    return total / n
