"""Fixtures for the non-default LovaszSoftmax options (losses/LovaszSoftmax.py:13-16,27-29,44-55,72-80 of the reference:
per_image, classes_to_ignore, classes_to_consider = 'all' / a list), generated with the REAL reference.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lovasz_opts.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402

R = ref_harness.load()
CASES = {
    "per_image": {"experiment": 3, "per_image": True},
    "ignore25": {"experiment": 3, "classes_to_ignore": 25},
    "all": {"experiment": 3, "classes_to_consider": "all"},
    "list": {"experiment": 3, "classes_to_consider": [0, 3, 7, 12, 24]},
    "per_image_ignore_list_e2": {"experiment": 2, "per_image": True, "classes_to_ignore": 17, "classes_to_consider": [1, 2, 5, 16]},
}
out = {}
for name, cfg in CASES.items():
    K = 25 if cfg["experiment"] == 3 else 17
    g = torch.Generator().manual_seed(len(name) * 7 + K)
    lg = (2 * torch.randn(3, K, 20, 28, generator=g)).requires_grad_()
    lb = torch.randint(0, K + 1, (3, 5, 7), generator=g).repeat_interleave(4, 1).repeat_interleave(4, 2)
    lb[lb == 3] = 1                       # class 3 absent everywhere; image 2 has few classes
    lb[2, :12] = 0
    loss = R.losses.LovaszSoftmax(dict(cfg))(lg, lb)
    loss.backward()
    out[name + ":logits"], out[name + ":labels"] = lg.detach().numpy().copy(), lb.numpy().copy()
    out[name + ":loss"], out[name + ":grad"] = np.array(float(loss)), lg.grad.numpy().copy()
    print(name, float(loss))
np.savez_compressed(os.path.join(HERE, "lovasz_options.npz"), **out)
print("wrote lovasz_options.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "lovasz_options.npz")) / 1024))
