"""Pins oracle.nets.deeplabv3_forward (models/DeepLabv3.py:58-141 of the reference) against the fixture generated
from the REAL reference at a size where the ASPP dilations 12 / 24 / 36 read in-image taps
(tests/golden/make_golden_deeplabv3.py).  CPU only."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_deeplabv3 import make_inputs, summarise  # noqa: E402  (input generator shared with the fixture script)

from oracle import losses as OL  # noqa: E402
from oracle import nets as ON  # noqa: E402
from oracle.state import fill_state  # noqa: E402


def test_deeplabv3_oracle_matches_reference_fixture(golden):
    g = golden("deeplabv3_r50_e2_d36")
    spec = json.loads(str(g["spec"]))
    S = fill_state(spec, int(g["seed"]))
    x, lbl = make_inputs()
    assert tuple(x.shape) == tuple(g["shape"])
    with torch.no_grad():
        e = ON.deeplabv3_forward({k: v.clone() for k, v in S.items()}, x, train=False)
    s = summarise(e)
    scale = float(g["eval_scale"])
    assert np.abs(s["sub"] - g["eval_sub"]).max() <= 1e-5 * scale
    assert np.abs(s["rows"] - g["eval_rows"]).max() <= 1e-5 * scale
    assert np.array_equal(e.argmax(1).numpy().astype(np.uint8), g["eval_argmax"])
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        S[k].requires_grad_()
    y = ON.deeplabv3_forward(S, x, train=True)
    loss = OL.lovasz_softmax(y, lbl)
    loss.backward()
    s = summarise(y)
    assert np.abs(s["sub"] - g["train_sub"]).max() <= 1e-5 * np.abs(g["train_sub"]).max()
    assert abs(float(loss) - float(g["losses"][0])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(S[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-9)
    for i in (2, 3, 4):   # the dilated branches' filter gradients, tap by tap
        ref = g["g:aspp.aspp%d.weight[0:2]" % i]
        np.testing.assert_allclose(S["aspp.aspp%d.weight" % i].grad[0:2].numpy(), ref, atol=1e-5 * np.abs(ref).max())
        assert np.abs(ref[:, :, 0, 0]).max() > 0 and np.abs(ref[:, :, 2, 2]).max() > 0   # off-centre taps are live
    for k in g.files:
        if k.startswith("rs:"):
            np.testing.assert_allclose(S[k[3:]].detach().numpy(), g[k], rtol=1e-5, atol=1e-6)
