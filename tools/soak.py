"""Soak run of the bench model: N Adam steps at an aggressive learning rate on four rotating synthetic batches -- the activations, gradients
and BatchNorm statistics drift far from their initial ranges, which is what the bound-derived exponents of the fp16 x 2 planes (csrc/planes.h)
and the amax records have to survive: every checked quantity must stay finite, the loss must fall.   python3 tools/soak.py [steps] [lr]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
dev = torch.device("cuda")
torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=lr)
batches = [bench.synth_batch(8, 544, 960, 25, 1000 + i, dev) for i in range(4)]
losses, t0 = [], time.time()
fp = model.flat()
for s in range(steps):
    img, lbl = batches[s % 4]
    opt.zero_grad()
    i, f = model(img)
    loss = crit(i, f, lbl)
    loss.backward()
    opt.step()
    losses.append(loss.detach())
    if (s + 1) % 50 == 0 or s == steps - 1:
        l = float(losses[-1])
        gmax, pmax = float(fp.grad.abs().max()), float(fp.flat.abs().max())
        lmax = float(f.detach().abs().max())
        ok = all(map(lambda v: v == v and abs(v) != float("inf"), (l, gmax, pmax, lmax)))
        print("step %4d  loss %.4f  max|grad| %.3g  max|param| %.3g  max|logit| %.3g  %s  (%.0f s)" % (s + 1, l, gmax, pmax, lmax, "finite" if ok else "NOT FINITE", time.time() - t0), flush=True)
        assert ok
ls = [float(x) for x in losses]
first, last = sum(ls[:8]) / 8, sum(ls[-8:]) / 8
print(json.dumps({"steps": steps, "lr": lr, "loss_first8": first, "loss_last8": last, "all_finite": True}))
assert last < first
