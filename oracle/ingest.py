"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's per-frame input transforms:
remap_mask (utils/utils.py:23-47), FlipNP (utils/transforms.py:222-240), PadNP (utils/transforms.py:8-20),
torchvision ToTensor / Normalize as wired at utils/utils.py:440-447.

Pinned against tests/golden/ingest.npz (remap_mask / FlipNP / PadNP of the REAL reference, make_golden_ingest.py).
torchvision is absent: ToTensor / Normalize follow its published semantics — uint8 HWC -> float32 CHW divided
by 255; (x - mean) / std channel-wise — parity unpinned for those two lines."""
import numpy as np


def remap_mask(mask, class_remapping, to_network=True):
    """utils/utils.py:23-47"""
    n = max(sum(len(v) for v in class_remapping.values()), int(mask.max()) + 1)
    table = np.full(n, 255, dtype=np.uint8)
    for key, val in class_remapping.items():
        for v in val:
            table[v] = key
    out = table[mask]
    if to_network:
        out[out == 255] = len(class_remapping) - 1
    return out


def flip(img, lbl, flags):
    """FlipNP: bit 1 = vertical (axis 0), bit 0 = horizontal (axis 1), the same for image and label"""
    if flags & 2:
        img, lbl = np.flip(img, 0), np.flip(lbl, 0)
    if flags & 1:
        img, lbl = np.flip(img, 1), np.flip(lbl, 1)
    return img.copy(), lbl.copy()


def pad(arr, ver=(2, 2), hor=(0, 0)):
    """PadNP(ver, hor, 'reflect')"""
    width = (ver, hor) + (((0, 0),) if arr.ndim == 3 else ())
    return np.pad(arr, pad_width=width, mode="reflect")


def to_tensor(img_u8, mean=None, std=None):
    """ToTensor (+ Normalize): uint8 HWC -> float32 CHW in [0, 1]"""
    x = np.ascontiguousarray(img_u8.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    if mean is not None:
        x = (x - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    return x.astype(np.float32)


def ingest(img_u8, lbl_u8, class_remapping, flags, ver=(2, 2), mean=None, std=None):
    """one frame, in the reference's order: remap -> flip -> pad -> ToTensor (-> Normalize)"""
    lbl = remap_mask(lbl_u8, class_remapping).astype(np.int32)
    img, lbl = flip(img_u8, lbl, int(flags))
    return to_tensor(pad(img, ver), mean, std), pad(lbl, ver).astype(np.int64)
