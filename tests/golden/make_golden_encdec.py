"""Fixture for EncDec(ResNet18 + UPerNet) generated with the REAL reference (UPerNet/EncDec from the
reference, torchvision trunk from the oracle's restatement).
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_encdec.py"""
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
torch.manual_seed(9)
model = R.models.EncDec({"encoder": {"model": "ResNet18", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 1)
spec = spec_of(model.state_dict())
model.load_state_dict(fill_state(spec, 400))
g = torch.Generator().manual_seed(401)
x = torch.rand(2, 3, 96, 128, generator=g)
lbl = torch.randint(0, 8, (2, 96, 128), generator=g)
out = {"x": x.numpy().copy(), "lbl": lbl.numpy().copy(), "seed": np.array(400), "spec": np.array(json.dumps(spec))}
model.eval()
model.get_features = False
with torch.no_grad():
    out["eval_final"] = model(x).numpy().copy()
model.train()
model.get_features = True
feat, y = model(x)
L = R.losses.LovaszSoftmax({"experiment": 1})
loss = L(y, lbl)
loss.backward()
out["train_final"], out["train_feat"] = y.detach().numpy().copy(), feat.detach().numpy().copy()
out["loss"] = np.array(float(loss))
names = [k for k, _ in model.named_parameters()]
out["grad_names"] = np.array(json.dumps(names))
out["grad_norms"] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
np.savez_compressed(os.path.join(HERE, "encdec_r18_upernet_e1_tiny.npz"), **out)
print("wrote encdec fixture", sum(p.numel() for p in model.parameters()), float(loss))
