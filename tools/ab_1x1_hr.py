"""within-run A/B on the HRNet-W48 train step (lr = 0): which 1x1 layers run the f16x2 kernels (ops.B3_1X1_MIN_DIM / _MIN_PROD / _MIN_ROWS).
The OCR head of this model sits at stride 4 (8 x 136 x 240 = 261 120 pixels): its 256 <-> 512 layers are 4 x the rows of the ResNet50 models'
on which the thresholds were set in round 3.   python3 tools/ab_1x1_hr.py [rounds]"""
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
def setup(dim, prod, rows):
    torch.cuda.synchronize()
    ops.release_workspaces()
    ops.release_b3_cache()
    ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS = dim, prod, rows
CONFIGS = [("baseline 512 / 512x1024 / 60k", 512, 512 * 1024, 60000), ("256 / 256x512 / 200k", 256, 256 * 512, 200000),
           ("256 / 256x256 / 200k", 256, 256 * 256, 200000)]
res = {c[0]: [] for c in CONFIGS}
for rnd in range(rounds):
    for name, dim, prod, rows in CONFIGS:
        setup(dim, prod, rows)
        res[name].append(timeit())
        print("round %d %-32s %.1f ms/step" % (rnd, name, res[name][-1]), flush=True)
for name, dim, prod, rows in CONFIGS:
    setup(dim, prod, rows)
    ops.PROFILE = []
    step(); torch.cuda.synchronize()
    agg = {}
    for k, fl, e0, e1 in ops.PROFILE:
        a = agg.setdefault(k, [0.0, 0]); a[0] += e0.elapsed_time(e1); a[1] += 1
    ops.PROFILE = None
    print(name, "  ".join("%s %.1f ms x%d" % (k, v[0], v[1]) for k, v in sorted(agg.items()) if k.split("_")[0] in ("fwd", "dgrad", "wgrad", "split3")))
for name, v in res.items():
    print("%-32s min %.1f median %.1f ms" % (name, min(v), sorted(v)[len(v) // 2]))
