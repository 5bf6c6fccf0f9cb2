"""Pins oracle/hrnet.py against the fixture generated from the reference's HRNetv2 (CPU only)."""
import json

import numpy as np
import torch

from oracle import losses as OL
from oracle.hrnet import hrnetv2_forward
from oracle.state import fill_state


def test_hrnetv2_oracle_matches_reference(golden):
    g = golden("hrnetv2_e3_tiny")
    spec = json.loads(str(g["spec"]))
    S = fill_state(spec, int(g["seed"]))
    x, lbl = torch.from_numpy(g["x"]), torch.from_numpy(g["lbl"])
    with torch.no_grad():
        e = hrnetv2_forward(S, x, train=False)
    np.testing.assert_allclose(e.numpy(), g["eval_final"], atol=1e-5)
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        S[k].requires_grad_()
    y = hrnetv2_forward(S, x, train=True)
    np.testing.assert_allclose(y.detach().numpy(), g["train_final"], atol=1e-5)
    loss = OL.cross_entropy(y, lbl, 3)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    loss.backward()
    names = json.loads(str(g["grad_names"]))
    assert names == params
    norms = np.array([float(S[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-8)
    for k in g.files:
        if k.startswith("rs:"):
            np.testing.assert_allclose(S[k[3:]].numpy(), g[k], atol=1e-6)


def test_hrnet_product_keys_and_w48_size(golden):
    from miccai2021_cataract_semantic_segmentation_amd.models import HRNetv2, OCRNet
    spec = json.loads(str(golden("hrnetv2_e3_tiny")["spec"]))
    m = HRNetv2({}, 3)
    sd = m.state_dict()
    assert [k for k, _ in spec] == list(sd.keys()) and all(tuple(s) == tuple(sd[k].shape) for k, s in spec)
    w48 = HRNetv2({"hrnet": {"width": 48, "stage1_width": 64, "modules": (1, 4, 3)}}, 3)
    assert sum(p.numel() for p in w48.parameters()) == 65863705     # published HRNetV2-W48 size (65.86 M)
    ocr = OCRNet({"backbone": "hrnet48", "pretrained": False}, 3)
    assert ocr.high_out_channels == 720 and ocr.out_stride == 4
