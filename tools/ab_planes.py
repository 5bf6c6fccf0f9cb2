"""within-run comparison of trunk routes on the HRNet-W48 train step (lr = 0: every step does the same work): the round-3 route (fp32 tensors,
split inside the convolution kernels) against producer-written planes for all / some widths.  usage: ab_planes.py [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda")
torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=0.0)
img, lbl = bench.synth_batch(8, 544, 960, 25, 1000, dev)
def step():
    opt.zero_grad(); i, f = model(img); loss = crit(i, f, lbl); loss.backward(); opt.step(); return loss
def timeit(n=8):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
CONFIGS = [("in-kernel split", False, (), True), ("planes all widths", True, (48, 96, 192, 384), True), ("planes 96/192/384", True, (96, 192, 384), True),
           ("planes 48", True, (48,), True), ("planes 192/384", True, (192, 384), True), ("planes 96+ unfused", True, (96, 192, 384), False)]
if len(sys.argv) > 2:
    CONFIGS = [c for c in CONFIGS if c[0] in sys.argv[2].split(",")]
res = {n: [] for n, *_ in CONFIGS}
for rnd in range(rounds):
    for name, pl, widths, fuse in CONFIGS:
        ops.PLANES, ops.PLANES_WIDTHS, ops.BN_BWD_FUSE = pl, widths, fuse
        res[name].append(timeit())
        print("round %d %-20s %.1f ms/step" % (rnd, name, res[name][-1]), flush=True)
for name, v in res.items():
    print("%-20s min %.1f median %.1f ms" % (name, min(v), sorted(v)[len(v) // 2]))
