"""How long does the HOST need to enqueue one train step of the bench model (no synchronisation inside the loop)?  If it is close to the GPU's
step time the step is launch-bound in phases and every host hiccup becomes a GPU gap.  python3 tools/host_time.py"""
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
    t = [time.perf_counter()]
    opt.zero_grad(); i, f = model(img); t.append(time.perf_counter())
    loss = crit(i, f, lbl); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]
for _ in range(3): step()
torch.cuda.synchronize()
rows = []
t0 = time.perf_counter()
for _ in range(6):
    rows.append(step())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for r in rows:
    print("host ms: forward %.1f  loss %.1f  backward %.1f  adam %.1f  | total %.1f" % (*r, sum(r)))
print("6 steps enqueued in %.1f ms (%.1f per step); GPU finished %.1f ms later; GPU-bound step time %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t1 - t0) / 6, 1e3 * (t2 - t1), 1e3 * (t2 - t0) / 6))
# ---- the replayed step: how long does ONE hipGraphLaunch of the captured step hold its host thread? (the margin against the GPU's step time)
from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
g = GraphedTrainStep(model, lambda o, l: crit(o[0], o[1], l), opt, img, lbl)
for _ in range(3):
    g(img, lbl)
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(8):
    a = time.perf_counter()
    g(img, lbl)
    hs.append(1e3 * (time.perf_counter() - a))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph replay: host ms per call " + " ".join("%.1f" % h for h in hs))
print("graph replay: 8 steps enqueued in %.1f ms (%.1f per step of host time); GPU-bound step time %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t1 - t0) / 8, 1e3 * (t2 - t0) / 8))
torch.cuda.synchronize()
hs = []
for _ in range(4):                      # with an idle GPU in front of every call: the pure enqueue cost of the graph
    torch.cuda.synchronize()
    a = time.perf_counter()
    g(img, lbl)
    hs.append(1e3 * (time.perf_counter() - a))
torch.cuda.synchronize()
print("graph replay on an idle GPU: host ms per call " + " ".join("%.1f" % h for h in hs))
