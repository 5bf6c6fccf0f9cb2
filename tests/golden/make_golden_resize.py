"""The resize-on-mismatch branch of the reference's losses (losses/TwoScaleLoss.py:45-48, losses/OhemCrossEntropy.py:23-26) from the REAL
reference: logits at a lower resolution than the labels, see make_golden.py.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_resize.py
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402

warnings.filterwarnings("ignore")
R = ref_harness.load()
out = {}
CASES = [
    # name, loss, experiment, K, logits (h, w) of interm, labels (H, W), batch
    ("ts_lovasz", "LovaszSoftmax", 3, 25, (6, 10), (24, 40), 2),        # x4, the OCRNet stride
    ("ts_ce", "CrossEntropyLoss", 2, 17, (5, 7), (18, 26), 2),          # non-integer ratio
    ("ts_ohem", "OhemCrossEntropy", 3, 25, (6, 6), (24, 24), 1),
]
for name, lname, exp, K, (h, w), (H, W), B in CASES:
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    interm = (torch.randn(B, K, h, w, generator=g) * 2).requires_grad_()
    final = (torch.randn(B, K, H, W, generator=g) * 2).requires_grad_()
    target = torch.randint(0, K + 1, (B, H, W), generator=g)
    extra = {"min_kept": 150, "thresh": 0.6} if lname == "OhemCrossEntropy" else {}
    cfg = {"experiment": exp, "interm": dict({"name": lname, "args": [], "weight": 0.4}, **extra),
           "final": dict({"name": lname, "args": [], "weight": 1.0}, **extra)}
    crit = R.losses.TwoScaleLoss(cfg)
    loss = crit(interm, final, target)
    loss.backward()
    out[name + "_interm"] = interm.detach().numpy().copy()
    out[name + "_final"] = final.detach().numpy().copy()
    out[name + "_target"] = target.numpy().copy()
    out[name + "_loss"] = np.float64(loss.item())
    out[name + "_ginterm"] = interm.grad.numpy().copy()
    out[name + "_gfinal"] = final.grad.numpy().copy()
    print(name, float(loss))
# OhemCrossEntropy called directly with low-resolution scores
g = torch.Generator().manual_seed(77)
score = (torch.randn(2, 17, 7, 9, generator=g) * 2).requires_grad_()
target = torch.randint(0, 18, (2, 26, 35), generator=g)
loss = R.losses.OhemCrossEntropy({"experiment": 2, "min_kept": 500, "thresh": 0.5})(score, target)
loss.backward()
out["ohem_score"], out["ohem_target"] = score.detach().numpy().copy(), target.numpy().copy()
out["ohem_loss"], out["ohem_grad"] = np.float64(loss.item()), score.grad.numpy().copy()
print("ohem", float(loss))
np.savez_compressed(os.path.join(HERE, "losses_resize.npz"), **out)
print("wrote losses_resize.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "losses_resize.npz")) / 1024))
