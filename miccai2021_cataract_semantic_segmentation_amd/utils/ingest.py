"""Raw CaDIS frames -> network input on the GPU (one HIP kernel per batch).

Mirror of what ``DatasetFromDF.__getitem__`` (datasets/Dataset_from_df.py:31-69 of the reference) does per frame on the
host: ``remap_mask(..., to_network=True)`` (utils/utils.py:23-47), ``FlipNP`` (utils/transforms.py:222-240),
``PadNP(ver=(2, 2), hor=(0, 0), 'reflect')`` (utils/transforms.py:8-20, wired at utils/utils.py:394-401), ``ToTensor`` and the
optional ``Normalize`` (utils/utils.py:440-447); optionally the PIL blur / colour jitter between pad and ToTensor (utils/augment.py).
The affine / crop augmentations stay on the host side of the boundary."""
import numpy as np
import torch

from .classes import CLASS_REMAP

TORCHVISION_MEAN, TORCHVISION_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]   # utils/utils.py:445-447


def remap_lut(experiment):
    """256-entry table equal to remap_mask(mask, CLASS_INFO[experiment][0], to_network=True) (utils/utils.py:23-47)"""
    remap = CLASS_REMAP[experiment]
    lut = np.full(256, 255, dtype=np.uint8)
    for key, raw in remap.items():
        for v in raw:
            lut[v] = key
    lut[lut == 255] = len(remap) - 1
    return lut


def sample_flips(batch, probability=(0.0, 0.5), random=np.random):
    """FlipNP's draws (utils/transforms.py:227-239): per frame one uniform for the vertical, then one for the horizontal
    flip.  Returns int32 flags, bit 0 = horizontal, bit 1 = vertical."""
    flags = np.zeros(batch, dtype=np.int32)
    for i in range(batch):
        if random.random() < probability[0]:
            flags[i] |= 2
        if random.random() < probability[1]:
            flags[i] |= 1
    return flags


class GpuIngest:
    """``ingest(img_u8 [B,H,W,3], lbl_u8 [B,H,W], flips) -> (x float32 [B,3,H',W], labels int64 [B,H',W])``"""

    def __init__(self, experiment, pad=(2, 2), normalise=False, device="cuda"):
        self.device = torch.device(device)
        self.pad = pad
        self.lut = torch.from_numpy(remap_lut(experiment)).to(self.device)
        self.mean = torch.tensor(TORCHVISION_MEAN, device=self.device) if normalise else None
        self.std = torch.tensor(TORCHVISION_STD, device=self.device) if normalise else None

    def __call__(self, img, lbl, flips=None, nhwc4=False, blur_radii=None, jitter=None):
        """blur_radii: per-frame GaussianBlur radius (0 = none; utils.augment.sample_blur); jitter: (orders, factors) of
        utils.augment.sample_color_jitter.  With either, the image takes the reference's order of operations on uint8
        (flip + reflect pad -> blur -> colour jitter -> ToTensor / Normalize: utils/utils.py:394-447) in a few more kernels."""
        img = img.to(self.device, non_blocking=True) if img is not None else None
        lbl = lbl.to(self.device, non_blocking=True) if lbl is not None else None
        if flips is not None:
            flips = torch.as_tensor(np.asarray(flips, dtype=np.int32)).to(self.device, non_blocking=True)
        from .. import ops   # needs libcatseg_hip.so; the table / flag helpers above do not
        if img is not None and (blur_radii is not None or jitter is not None):
            from .augment import GpuAugment
            aug = GpuAugment(self.device)
            u8 = ops.aug_pad_flip_u8(img.contiguous(), flips, self.pad[0], self.pad[1])
            if blur_radii is not None:
                u8 = aug.blur(u8, blur_radii)
            if jitter is not None:
                u8 = aug.color_jitter(u8, jitter[0], jitter[1])
            x = ops.ingest_u8(u8, None, self.lut, None, 0, 0, self.mean, self.std, nhwc4=nhwc4)[0]
            labels = ops.ingest_u8(None, lbl, self.lut, flips, self.pad[0], self.pad[1], self.mean, self.std)[1] if lbl is not None else None
            return x, labels
        return ops.ingest_u8(img, lbl, self.lut, flips, self.pad[0], self.pad[1], self.mean, self.std, nhwc4=nhwc4)
