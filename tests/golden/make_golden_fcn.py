"""Fixture for the reference's FCN-8s (models/FCN.py:7-61: conv + ReLU, 2x2 max-pooling, ConvTranspose2d up-sampling), generated with the
REAL reference:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_fcn.py

width 0.25 (16 .. 256 channels), task 2 (17 classes), input 2 x 3 x 96 x 128 (pool5: 3 x 4).  Inputs / labels / weights are regenerated
from seeds; the fixture stores the logits of the first step (every 4th row and column, three full rows, sums), the loss of two Adam
steps, and per-parameter gradient norms / sums plus the full gradients of the three ConvTranspose2d layers' biases and of one filter of each.
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

SEED, SHAPE, EXP, WIDTH = 700, (2, 3, 96, 128), 2, 0.25


def make_inputs(seed=SEED, shape=SHAPE, K=17):
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(shape, generator=g)
    lbl = torch.randint(0, K + 1, (shape[0], shape[2] // 16, shape[3] // 16), generator=g)
    lbl = lbl.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
    return x, lbl


def summarise(t):
    t = t.detach()
    return {"sub": t[:, :, ::4, ::4].numpy().copy(), "rows": t[:, :, [0, 47, 95], :].numpy().copy(),
            "sum": np.array(float(t.double().sum())), "abs": np.array(float(t.double().abs().sum()))}


if __name__ == "__main__":
    torch.set_num_threads(8)
    R = ref_harness.load()
    torch.manual_seed(SEED)
    model = R.models.FCN({"width": WIDTH}, EXP)
    spec = spec_of(model.state_dict())
    model.load_state_dict(fill_state(spec, SEED))
    x, lbl = make_inputs()
    out = {"seed": np.array(SEED), "shape": np.array(SHAPE), "spec": np.array(json.dumps(spec))}
    model.train()
    L = R.losses.LovaszSoftmax({"experiment": EXP})
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for s in range(2):
        opt.zero_grad()
        y = model(x)
        loss = L(y, lbl)
        loss.backward()
        if s == 0:
            for k, v in summarise(y).items():
                out["train_" + k] = v
            out["train_scale"] = np.array(float(y.abs().max()))
            names = [k for k, _ in model.named_parameters()]
            out["grad_names"] = np.array(json.dumps(names))
            out["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
            out["grad_sums"] = np.array([float(p.grad.double().sum()) for _, p in model.named_parameters()])
            for k, p in model.named_parameters():
                if k.startswith("deconv") and k.endswith(".bias"):
                    out["g:" + k] = p.grad.numpy().copy()
            for n in ("deconv32", "deconv16", "deconv8"):
                out["g:%s.weight[3]" % n] = getattr(model, n).weight.grad[3].numpy().copy()
            out["g:conv1.weight"] = model.conv1.weight.grad.numpy().copy()
        opt.step()
        losses.append(float(loss))
    out["losses"] = np.array(losses)
    path = os.path.join(HERE, "fcn_w025_e2.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KB; losses %s; scale %.4f" % (path, os.path.getsize(path) / 1024, losses, float(out["train_scale"])))
