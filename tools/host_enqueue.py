"""host-side cost of one training step: time to ENQUEUE a step (no synchronisation inside) against its GPU time, plus a cProfile of the
enqueue path.  usage: host_enqueue.py [--profile]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
dev = torch.device("cuda")
torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=1e-4)
img, lbl = bench.synth_batch(8, 544, 960, 25, 1000, dev)
def step():
    opt.zero_grad(); i, f = model(img); loss = crit(i, f, lbl); loss.backward(); opt.step(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
hs, gs = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); gs.append((t2 - t0) * 1e3)
print("host enqueue ms per step:", [round(h, 1) for h in hs], " step incl. GPU ms:", [round(g, 1) for g in gs])
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
