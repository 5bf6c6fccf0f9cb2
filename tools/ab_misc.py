"""within-run A/B of the persistent-block count of the planes kernel on the HRNet-W48 train step (lr = 0).  (Round 5 also tried the first branch
stream at high priority here: no gain, the knob left the product in round 6.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import engine, ops
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
def setup(prio, slots):
    torch.cuda.synchronize()
    engine._side_streams.clear()
    ops.release_workspaces()
    ops._bn_part.clear()
    ops.lib.catseg_debug_set_dconv3_pl_slots(slots)
CONFIGS = [("baseline", 0, 512), ("pl slots 384", 0, 384), ("pl slots 448", 0, 448), ("pl slots 640", 0, 640)]
res = {n: [] for n, *_ in CONFIGS}
for rnd in range(rounds):
    for name, prio, slots in CONFIGS:
        setup(prio, slots)
        res[name].append(timeit())
        print("round %d %-24s %.1f ms/step" % (rnd, name, res[name][-1]), flush=True)
for name, v in res.items():
    print("%-24s min %.1f median %.1f ms" % (name, min(v), sorted(v)[len(v) // 2]))
