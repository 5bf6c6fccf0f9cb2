"""within-run A/B of the trunk arithmetic (ops.TRUNK: f16x2 = two fp16 planes / three products in the direct forward / backward-data
kernels, bf16x3 = three planes / six products) on the HRNet-W48 train step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
dev = torch.device("cuda")
torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=0.0)
img, lbl = bench.synth_batch(8, 544, 960, 25, 1000, dev)
def step():
    opt.zero_grad(); i, f = model(img); loss = crit(i, f, lbl); loss.backward(); opt.step(); return loss
def timeit(n=8):
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    for trunk in ("f16x2", "bf16x3"):
        ops.TRUNK = trunk
        ops.PROFILE = []
        step(); torch.cuda.synchronize()
        kinds = {}
        for q in ops.PROFILE:
            kinds[q[0]] = kinds.get(q[0], 0) + 1
        ops.PROFILE = None
        print("round %d TRUNK=%s: %.1f ms/step (loss %.7f)  launches %s" % (rnd, trunk, timeit(), float(step()),
              {k: v for k, v in kinds.items() if "d3" in k}), flush=True)
