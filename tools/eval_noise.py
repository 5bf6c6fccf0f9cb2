"""fp32 rounding-noise calibration of the eval-mode logits of the OCRNet-R50 fixture: distance of the reference's own fp32
logits, the 3-kernel HIP path and the fused HIP inference path from an fp64 evaluation of the oracle (test tolerances
in tests/test_nets_gpu.py are set from this).  Run on a GPU box: python tools/eval_noise.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets as ON                                              # noqa: E402
from oracle.state import fill_state                                        # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd import engine           # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet    # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "ocrnet_r50_e3_tiny.npz"), allow_pickle=False)
spec = json.loads(str(g["spec"]))
S = fill_state(spec, int(g["seed"]))
x = torch.from_numpy(g["x"])
S64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in S.items()}
with torch.no_grad():
    f64 = ON.ocrnet_forward(S64, x.double(), train=False)[1].numpy()
    c32 = ON.ocrnet_forward(S, x, train=False)[1].double().numpy()
ref = g["eval_final"].astype(np.float64)
scale = np.abs(f64).max()
model = OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 3)
model.load_state_dict(S)
model.cuda().eval()
res = {}
with torch.no_grad():
    for fuse in (False, True):
        engine.FUSE_EVAL_BN = fuse
        res[fuse] = model(x.cuda())[1].double().cpu().numpy()


def d(a, b):
    return np.abs(a - b).max() / scale


print("scale %.1f" % scale)
print("reference fp32 vs f64   %.2e   oracle cpu fp32 vs f64 %.2e" % (d(ref, f64), d(c32, f64)))
print("hip 3-kernel  vs f64    %.2e   vs reference %.2e" % (d(res[False], f64), d(res[False], ref)))
print("hip fused     vs f64    %.2e   vs reference %.2e" % (d(res[True], f64), d(res[True], ref)))
print("hip fused vs 3-kernel   %.2e" % d(res[True], res[False]))
