"""Times one FCN-8s (models/FCN.py of the reference, width 1, 25 classes) train step at 2 x 3 x 512 x 960 on the HIP engine and prints the
per-kind kernel time -- a side figure for DESIGN.md (FCN is not the bench model).  python3 tools/time_fcn.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miccai2021_cataract_semantic_segmentation_amd import ops  # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.losses import LovaszSoftmax  # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.models import FCN  # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam  # noqa: E402

torch.manual_seed(0)
model = FCN({"width": 1}, 3).cuda().train()
opt = FusedAdam(model, lr=1e-4)
crit = LovaszSoftmax({"experiment": 3})
x = torch.rand(2, 3, 512, 960, device="cuda")
lbl = torch.randint(0, 26, (2, 16, 30), device="cuda").repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()


def step():
    opt.zero_grad()
    loss = crit(model(x), lbl)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    l = step()
torch.cuda.synchronize()
print("FCN-8s width 1, 2 x 3 x 512 x 960: %.2f ms / train step, loss %.4f" % ((time.perf_counter() - t0) * 100, float(l)))
ops.PROFILE = []
step()
torch.cuda.synchronize()
agg = {}
for k, fl, e0, e1 in ops.PROFILE:
    a = agg.setdefault(k, [0.0, 0.0, 0])
    a[0] += e0.elapsed_time(e1); a[1] += fl; a[2] += 1
ops.PROFILE = None
for k, (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("  %-12s %3d launches %8.3f ms %8.1f TFLOP/s" % (k, n, ms, fl / ms / 1e9 if ms > 0 else 0.0))
