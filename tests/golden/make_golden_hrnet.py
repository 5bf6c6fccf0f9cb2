"""Fixture for the reference's HRNetv2 (as shipped: widths 32/64/128/256), generated with the REAL reference.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hrnet.py"""
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

R = ref_harness.load()
torch.manual_seed(5)
model = R.models.HRNetv2({}, 3)
spec = spec_of(model.state_dict())
model.load_state_dict(fill_state(spec, 300))
g = torch.Generator().manual_seed(301)
x = torch.rand(2, 3, 64, 96, generator=g)
lbl = torch.randint(0, 26, (2, 64, 96), generator=g)
lbl[:, :20, :40] = 0
out = {"x": x.numpy().copy(), "lbl": lbl.numpy().copy(), "seed": np.array(300), "spec": np.array(json.dumps(spec))}
model.eval()
with torch.no_grad():
    out["eval_final"] = model(x).numpy().copy()
model.train()
ce = torch.nn.CrossEntropyLoss(ignore_index=25)
y = model(x)
loss = ce(y, lbl)
loss.backward()
out["train_final"] = y.detach().numpy().copy()
out["loss"] = np.array(float(loss))
names = [k for k, _ in model.named_parameters()]
out["grad_names"] = np.array(json.dumps(names))
out["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
sd = model.state_dict()
for k in ("bn1.running_mean", "stage4.0.fuse_layers.3.0.2.1.running_var"):
    out["rs:" + k] = sd[k].numpy().copy()
np.savez_compressed(os.path.join(HERE, "hrnetv2_e3_tiny.npz"), **out)
print("wrote hrnetv2_e3_tiny", sum(p.numel() for p in model.parameters()), "params", float(loss))
