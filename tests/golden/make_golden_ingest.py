"""Input-transform fixtures from the REAL reference (utils.remap_mask, utils.transforms.FlipNP / PadNP), see make_golden.py.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ingest.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402

R = ref_harness.load()
from utils import CLASS_INFO, remap_mask          # noqa: E402  (the reference's utils package)
from utils.transforms import FlipNP, PadNP        # noqa: E402

rng = np.random.RandomState(11)
B, H, W = 6, 10, 14
img = rng.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
lbl = rng.randint(0, 36, (B, H, W)).astype(np.uint8)
out = {"img": img, "lbl": lbl}
for exp in (1, 2, 3):
    np.random.seed(5 + exp)
    flipper, padder = FlipNP(probability=(0.4, 0.5)), PadNP(ver=(2, 2), hor=(0, 0), padding_mode="reflect")
    imgs, lbls, flags = [], [], []
    for b in range(B):
        m = remap_mask(lbl[b], CLASS_INFO[exp][0], to_network=True).astype("int32")
        i2, l2, meta = flipper((img[b], m, {}))
        fd = int(meta["flip_dims"])            # -1 horizontal, -2 vertical, -3 both (utils/transforms.py:239)
        flags.append({0: 0, -1: 1, -2: 2, -3: 3}[fd])
        imgs.append(padder(i2))
        lbls.append(padder(l2))
    out["e%d_img" % exp] = np.stack(imgs)
    out["e%d_lbl" % exp] = np.stack(lbls).astype(np.int64)
    out["e%d_flags" % exp] = np.array(flags, dtype=np.int32)
    lut_in = np.arange(36, dtype=np.uint8)
    out["e%d_lut36" % exp] = remap_mask(lut_in, CLASS_INFO[exp][0], to_network=True)
    print(exp, flags)
np.savez_compressed(os.path.join(HERE, "ingest.npz"), **out)
print("wrote ingest.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "ingest.npz")) / 1024))
