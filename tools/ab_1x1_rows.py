"""within-run A/B: minimum number of pixels for wide 1x1 layers on the split-precision kernels (ops.B3_1X1_MIN_ROWS) -- OCRNet-ResNet50 /
DeepLabv3+ train step (their stride-8 maps have 65 280 pixels at bs 8: below the default 131 072)"""
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
model = OCRNet(dict(bench.MODELS["ocrnet_r50"][0]), 3).to(dev).train()
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
for rnd in range(2):
    for rows, dim, prod in ((131072, 512, 512 * 1024), (60000, 512, 512 * 1024), (60000, 256, 256 * 1024), (60000, 256, 256 * 512)):
        ops.B3_1X1_MIN_ROWS, ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD = rows, dim, prod
        ops.release_b3_cache()
        ops.PROFILE = []
        step(); torch.cuda.synchronize()
        n = sum(1 for q in ops.PROFILE if q[0].endswith("_h2"))
        ops.PROFILE = None
        print("round %d min rows %d dim %d product %d: %d f16x2 launches, %.1f ms/step (loss %.7f)" % (rnd, rows, dim, prod, n, timeit(), float(step())), flush=True)
