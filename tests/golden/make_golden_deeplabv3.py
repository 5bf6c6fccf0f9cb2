"""Fixture for the reference's DeepLabv3 (configs/DeepLabv3_rf_lvsz.json: ResNet50, output stride 8, task 2, LovaszSoftmax),
generated with the REAL reference (models/DeepLabv3.py:11-141; torchvision trunk from the oracle's restatement).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_deeplabv3.py

The input is 2 x 3 x 304 x 320 so that the stride-8 feature map is 38 x 40: the ASPP branches with dilation 12 / 24 / 36
(mult = 2 at output stride 8, models/DeepLabv3.py:45) all read IN-IMAGE off-centre taps (on the 8 x 12 map of the tiny
fixtures every off-centre tap of d = 24 / 36 falls into the padding and the 3x3 degenerates to a 1x1).
To keep the fixture small, inputs / labels / weights are regenerated from seeds (torch CPU generators are deterministic),
and the 2 x 17 x 304 x 320 outputs are stored as: every 8th row and column, three full rows, the full argmax map and the
top-2 margin map (float16) of the eval logits.
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

SEED, SHAPE, EXP = 500, (2, 3, 304, 320), 2


def make_inputs(seed=SEED, shape=SHAPE, K=17):
    """shared with the tests: images uniform [0,1), blob-structured labels incl. the ignore label K"""
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(shape, generator=g)
    lbl = torch.randint(0, K + 1, (shape[0], shape[2] // 16, shape[3] // 16), generator=g)
    lbl[lbl == 3] = 4
    lbl[lbl == 11] = 0
    lbl = lbl.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
    return x, lbl


def summarise(t):
    """compact evidence of a B x K x H x W logits tensor"""
    t = t.detach()
    return {"sub": t[:, :, ::8, ::8].numpy().copy(), "rows": t[:, :, [0, 151, 303], :].numpy().copy(),
            "sum": np.array(float(t.double().sum())), "abs": np.array(float(t.double().abs().sum()))}


if __name__ == "__main__":
    torch.set_num_threads(8)
    R = ref_harness.load()
    torch.manual_seed(SEED)
    model = R.models.DeepLabv3({"backbone": "resnet50", "aspp": {"channels": 256}, "out_stride": 8, "pretrained": False}, EXP)
    spec = spec_of(model.state_dict())
    model.load_state_dict(fill_state(spec, SEED))
    x, lbl = make_inputs()
    out = {"seed": np.array(SEED), "shape": np.array(SHAPE), "spec": np.array(json.dumps(spec))}
    model.eval()
    with torch.no_grad():
        e = model(x)
    for k, v in summarise(e).items():
        out["eval_" + k] = v
    out["eval_argmax"] = e.argmax(1).numpy().astype(np.uint8)
    top2 = e.topk(2, dim=1).values
    out["eval_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    out["eval_scale"] = np.array(float(e.abs().max()))
    model.train()
    L = R.losses.LovaszSoftmax({"experiment": EXP})
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    losses = []
    for s in range(2):
        opt.zero_grad()
        y = model(x)
        loss = L(y, lbl)
        loss.backward()
        if s == 0:
            for k, v in summarise(y).items():
                out["train_" + k] = v
            names = [k for k, _ in model.named_parameters()]
            out["grad_names"] = np.array(json.dumps(names))
            out["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
            out["grad_sums"] = np.array([float(p.grad.double().sum()) for _, p in model.named_parameters()])
            for k, p in model.named_parameters():
                if k.startswith("aspp.") and p.numel() <= 4096:
                    out["g:" + k] = p.grad.numpy().copy()
            # one filter row of each dilated branch: the taps a wrong dilation / tap order would scramble
            for i in (2, 3, 4):
                out["g:aspp.aspp%d.weight[0:2]" % i] = model.aspp.__getattr__("aspp%d" % i).weight.grad[0:2].numpy().copy()
            sd = model.state_dict()
            for k in ("aspp.aspp4_bn.running_mean", "aspp.aspp4_bn.running_var", "aspp.bn2.running_var"):
                out["rs:" + k] = sd[k].numpy().copy()
        opt.step()
        losses.append(float(loss))
    out["losses"] = np.array(losses)
    path = os.path.join(HERE, "deeplabv3_r50_e2_d36.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KB; losses %s; eval scale %.3f" % (path, os.path.getsize(path) / 1024, losses, float(out["eval_scale"])))
