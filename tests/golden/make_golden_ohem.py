"""OhemCrossEntropy fixtures from the REAL reference (losses/OhemCrossEntropy.py), see make_golden.py.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ohem.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402

R = ref_harness.load()
out = {}
CASES = [
    # name, experiment, K, shape, config extras, logit scale
    ("a", 3, 25, (1, 16, 24), {}, 3.0),                                   # min_kept (1e5) > n: k = n - 1 -> threshold = max p
    ("b", 3, 25, (1, 16, 24), {"min_kept": 100, "thresh": 0.7}, 3.0),     # thresh dominates
    ("c", 2, 17, (2, 12, 16), {"min_kept": 250, "thresh": 0.001}, 2.0),    # the k-th smallest probability dominates
    ("d", 1, 8, (1, 16, 16), {"min_kept": 10, "thresh": 0.3}, 1.0),       # experiment 1: nothing ignored
]
for name, exp, K, (B, H, W), extra, scale in CASES:
    g = torch.Generator().manual_seed(100 + ord(name))
    logits = (torch.randn(B, K, H, W, generator=g) * scale).requires_grad_()
    hi = K + 1 if exp in (2, 3) else K
    target = torch.randint(0, hi, (B, H, W), generator=g)
    cfg = {"experiment": exp}
    cfg.update(extra)
    crit = R.losses.OhemCrossEntropy(cfg)
    loss = crit(logits, target)
    loss.backward()
    out[name + "_logits"] = logits.detach().numpy().copy()
    out[name + "_target"] = target.numpy().copy()
    out[name + "_loss"] = np.float64(loss.item())
    out[name + "_grad"] = logits.grad.numpy().copy()
    out[name + "_cfg"] = np.array([exp, extra.get("min_kept", -1), extra.get("thresh", -1.0)], dtype=np.float64)
    print(name, float(loss), int((logits.grad.abs().sum(1) > 0).sum()), "pixels selected of", target.numel())
np.savez_compressed(os.path.join(HERE, "ohem.npz"), **out)
print("wrote ohem.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "ohem.npz")) / 1024))
