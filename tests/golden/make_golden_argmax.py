"""Label-map fixtures for the bit-exact argmax requirement (BASELINE north star: "bit-exact for argmax label maps"),
generated with the REAL reference networks in eval mode.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_argmax.py

An argmax map can only be required to be bit-identical between two fp32 implementations where the top-2 logit margin
exceeds the fp32 rounding noise (~1e-5 of the logit scale on these nets; a re-ordered summation moves a logit by that much
on any hardware, the reference on another CPU included).  For each net the input seed is therefore chosen, among 256
candidates (4 weight seeds x 64 image seeds), as the one whose reference logits have the LARGEST minimum top-2 margin over all pixels while the label map
still shows >= 3 classes; the fixture stores those seeds, the reference's two largest logits per pixel, the label map and the minimum margin, and
tests/test_argmax_gpu.py asserts torch.equal on the whole map for both eval paths (separate BN kernels and folded BN).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402
from oracle.state import fill_state, spec_of  # noqa: E402

SHAPE = (2, 3, 64, 96)
NETS = {
    "ocrnet_r50": ("OCRNet", {"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 3, 610),
    "deeplabv3plus_r50": ("DeepLabv3Plus", {"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 2, 620),
    "deeplabv3_r50": ("DeepLabv3", {"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 2, 630),
    "hrnetv2": ("HRNetv2", {}, 3, 640),
}


# OCRNet in eval mode with fill_state's arbitrary running statistics is badly conditioned (logits of scale 300-500 through two
# peaked softmaxes: any two fp32 implementations differ by ~1e-3 of the scale there).  "Trained-like" statistics: one
# training-mode forward with BatchNorm momentum 1 sets every running mean / variance to the batch statistics of a calibration
# batch; those statistics are stored in the fixture.
CALIBRATE = ("ocrnet_r50",)


def calibrate_bn(model):
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    old = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(image(6999))
    for m, o in zip(bns, old):
        m.momentum = o


def image(seed):
    return torch.rand(SHAPE, generator=torch.Generator().manual_seed(seed))


if __name__ == "__main__":
    torch.set_num_threads(8)
    R = ref_harness.load()
    out = {}
    for name, (cls, cfg, exp, wseed) in NETS.items():
        best = None
        for ws in range(wseed, wseed + 4):           # weight seeds: some random nets predict a single class everywhere
            torch.manual_seed(ws)
            model = getattr(R.models, cls)(dict(cfg), exp)
            spec = spec_of(model.state_dict())
            model.load_state_dict(fill_state(spec, ws))
            if name in CALIBRATE:
                calibrate_bn(model)
            model.eval()
            if hasattr(model, "get_intermediate"):
                model.get_intermediate = False
            for seed in range(64):
                with torch.no_grad():
                    o = model(image(7000 + seed))
                t = o.topk(2, dim=1).values
                rel = float((t[:, 0] - t[:, 1]).min()) / float(o.abs().max())
                ncls = len(o.argmax(1).unique())
                if ncls >= 3 and (best is None or rel > best[0]):
                    best = (rel, 7000 + seed, o, ncls, ws)
                    stats = {k: v.numpy().copy() for k, v in model.state_dict().items() if "running_" in k} if name in CALIBRATE else {}
        wseed = best[4]
        best = best[:4]
        for k, v in stats.items():
            out[name + ":rs:" + k] = v
        rel, seed, o, ncls = best
        print("%s: image seed %d, min top-2 margin %.3g of the logit scale %.2f, %d classes in the map" % (name, seed, rel, float(o.abs().max()), ncls))
        out[name + ":spec"] = np.array(json.dumps(spec))
        out[name + ":wseed"], out[name + ":xseed"] = np.array(wseed), np.array(seed)
        out[name + ":top2"] = o.topk(2, dim=1).values.numpy().copy()     # [B, 2, H, W]: the two largest logits per pixel
        out[name + ":scale"] = np.array(float(o.abs().max()))
        out[name + ":argmax"] = o.argmax(1).numpy().astype(np.uint8)
        out[name + ":min_margin_rel"] = np.array(rel)
    np.savez_compressed(os.path.join(HERE, "argmax_maps.npz"), **out)
    print("wrote argmax_maps.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "argmax_maps.npz")) / 1024))
