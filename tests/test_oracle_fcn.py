"""Pins oracle.nets.fcn_forward (models/FCN.py:40-61 of the reference) against the fixture generated from the REAL reference
(tests/golden/make_golden_fcn.py).  CPU only."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden_fcn import make_inputs, summarise  # noqa: E402  (input generator shared with the fixture script)

from oracle import losses as OL  # noqa: E402
from oracle import nets as ON  # noqa: E402
from oracle.state import fill_state  # noqa: E402


def test_fcn_oracle_matches_reference_fixture(golden):
    g = golden("fcn_w025_e2")
    spec = json.loads(str(g["spec"]))
    S = fill_state(spec, int(g["seed"]))
    x, lbl = make_inputs()
    assert tuple(x.shape) == tuple(g["shape"])
    for v in S.values():
        v.requires_grad_()
    y = ON.fcn_forward(S, x)
    loss = OL.lovasz_softmax(y, lbl)
    loss.backward()
    s = summarise(y)
    scale = float(g["train_scale"])
    assert np.abs(s["sub"] - g["train_sub"]).max() <= 1e-6 * scale
    assert np.abs(s["rows"] - g["train_rows"]).max() <= 1e-6 * scale
    assert abs(float(loss) - float(g["losses"][0])) < 1e-6
    names = json.loads(str(g["grad_names"]))
    norms = np.array([float(S[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-10)
    for k in g.files:
        if k.startswith("g:") and k.endswith("[3]"):
            ref = g[k]
            np.testing.assert_allclose(S[k[2:-3]].grad[3].numpy(), ref, atol=1e-5 * np.abs(ref).max())
            assert np.abs(ref[:, 0, 0]).max() > 0 and np.abs(ref[:, -1, -1]).max() > 0     # corner taps of the transposed filters are live
        elif k.startswith("g:"):
            np.testing.assert_allclose(S[k[2:]].grad.numpy(), g[k], atol=1e-5 * np.abs(g[k]).max())
