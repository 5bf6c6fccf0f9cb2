"""diagnostic: per-branch ASPP filter-gradient error of the DeepLabv3 fixture under fp32 / bf16x3 with and without the direct kernels"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_deeplabv3 import make_inputs
from oracle.state import fill_state
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd.models import DeepLabv3
from miccai2021_cataract_semantic_segmentation_amd.losses import LovaszSoftmax
g = np.load(os.path.join(ROOT, "tests/golden/deeplabv3_r50_e2_d36.npz"))
spec = json.loads(str(g["spec"]))
x, lbl = make_inputs(); xd, ld = x.cuda(), lbl.cuda()
for prec, d3 in (("fp32", False), ("bf16x3", False), ("bf16x3", True)):
    ops.PRECISION = prec; ops.DCONV3 = d3
    if prec == "bf16x3":
        ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = 1, 64, 32, 1, 1
        ops.DCONV3_MIN_ROWS = 1
    model = DeepLabv3({"backbone": "resnet50", "aspp": {"channels": 256}, "out_stride": 8, "pretrained": False}, 2)
    model.load_state_dict(fill_state(spec, int(g["seed"]))); model.cuda().train()
    y = model(xd); loss = LovaszSoftmax({"experiment": 2})(y, ld); loss.backward()
    P = dict(model.named_parameters())
    out = []
    for i in (2, 3, 4):
        ref = g["g:aspp.aspp%d.weight[0:2]" % i]; got = P["aspp.aspp%d.weight" % i].grad[0:2].cpu().numpy()
        out.append("%.3f%% (L2 %.3f%%)" % (100 * np.abs(got - ref).max() / np.abs(ref).max(),
                                          100 * np.linalg.norm((got - ref).ravel().astype(np.float64)) / np.linalg.norm(ref.ravel().astype(np.float64))))
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(P[k].grad.double().norm()) for k in names])
    print(prec, "d3" if d3 else "no-d3", "loss %.7f (ref %.7f)" % (float(loss), float(g["losses"][0])), "aspp wgrad err", out,
          "max grad-norm rel err %.3g" % np.abs(norms / g["grad_norms"] - 1).max(), flush=True)
    ops.release_b3_cache()
