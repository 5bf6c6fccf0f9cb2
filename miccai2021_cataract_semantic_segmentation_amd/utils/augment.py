"""The PIL-based training augmentations of the reference's image pipeline, on the GPU and on uint8 batches.

Mirror of ``BlurPIL(probability=.05, kernel_limits=(3, 7))`` (utils/transforms.py:242-251) and torchvision's
``ColorJitter(brightness=(2/3, 1.5), contrast=(2/3, 1.5), saturation=(2/3, 1.5), hue=(-.05, .05))`` as the reference wires them
(utils/utils.py:412-417: after ``PadNP`` / ``FlipNP`` and ``ToPILImage``, before ``ToTensor``).  The kernels (csrc/augment.hip)
reproduce Pillow's integer / float arithmetic bit for bit (tests/golden/augment.npz is generated with Pillow);
the random draws happen on the host: ``sample_blur`` follows ``BlurPIL.__call__``'s numpy draws, ``sample_color_jitter`` the
order of torchvision >= 0.9 ``ColorJitter.get_params`` (torchvision is not installed here: that order is unpinned)."""
import math

import numpy as np
import torch

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def gaussian_box_params(radius):
    """(int radius, ww, fw) of libImaging/BoxBlur.c for ImageFilter.GaussianBlur(radius): three box passes per direction"""
    sigma2 = float(radius * radius) / 3
    L = math.sqrt(12.0 * sigma2 + 1.0)
    l = math.floor((L - 1.0) / 2.0)
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2)
    a /= 6 * (sigma2 - (l + 1) * (l + 1))
    fr = np.float32(l + a)
    r = int(fr)
    ww = int(np.float32(1 << 24) / (fr * np.float32(2) + np.float32(1)))        # (UINT32)(1 << 24) / (floatRadius * 2 + 1): float arithmetic
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww & 0xFFFFFFFF, fw & 0xFFFFFFFF


def sample_blur(batch, probability=0.05, kernel_limits=(3, 7), random=np.random):
    """BlurPIL.__call__'s draws per frame: one uniform, then (if it fires) np.random.randint(*kernel_limits); 0 = no blur"""
    radii = np.zeros(batch, dtype=np.int32)
    for i in range(batch):
        if random.random() < probability:
            radii[i] = random.randint(*kernel_limits)
    return radii


def sample_color_jitter(batch, brightness=(2 / 3, 1.5), contrast=(2 / 3, 1.5), saturation=(2 / 3, 1.5), hue=(-0.05, 0.05), generator=None):
    """per frame: a permutation of the four operations, then one uniform factor each in the order brightness, contrast,
    saturation, hue (torchvision >= 0.9 ColorJitter.get_params).  Returns (orders int32 [B, 4], factors float64 [B, 4])"""
    orders = np.zeros((batch, 4), dtype=np.int32)
    factors = np.zeros((batch, 4), dtype=np.float64)
    for i in range(batch):
        orders[i] = torch.randperm(4, generator=generator).numpy()
        for k, (lo, hi) in enumerate((brightness, contrast, saturation, hue)):
            factors[i, k] = float(torch.empty(1).uniform_(lo, hi, generator=generator))
    return orders, factors


class GpuAugment:
    """``blur(img_u8 [B,H,W,3], radii)`` and ``color_jitter(img_u8, orders, factors)`` on device tensors; both return new tensors"""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)

    def _dev(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device, non_blocking=True)

    def blur(self, img, radii):
        from .. import ops
        radii = np.asarray(radii)
        if not (radii > 0).any():
            return img
        prm = np.array([gaussian_box_params(int(r)) if r > 0 else (-1, 0, 0) for r in radii], dtype=np.int64)
        return ops.aug_gaussian_blur(img, self._dev(prm[:, 0], torch.int32), self._dev(prm[:, 1].astype(np.uint32).view(np.int32), torch.int32),
                                     self._dev(prm[:, 2].astype(np.uint32).view(np.int32), torch.int32))

    def color_jitter(self, img, orders, factors):
        from .. import ops
        orders, factors = np.asarray(orders), np.asarray(factors, dtype=np.float64)
        out = img.clone()
        for step in range(orders.shape[1]):
            op = orders[:, step].astype(np.int32)
            f = np.array([factors[b, o] if o >= 0 else 0.0 for b, o in enumerate(op)], dtype=np.float64)
            hue = op == HUE
            f[hue] = [float(int(v * 255) & 0xFF) for v in f[hue]]      # np.uint8(hue_factor * 255): truncation in double, uint8 wrap-around
            ops.aug_color_op(out, self._dev(op, torch.int32), self._dev(f.astype(np.float32), torch.float32))
        return out
