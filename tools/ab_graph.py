"""Eager launch loop against the whole step replayed as one hipGraph (graph.GraphedTrainStep) on the bench workload: GPU step time (barrier to
barrier), HOST time per step (time until the step's launches are enqueued), bit-identity of the parameters after the same steps.
    python3 tools/ab_graph.py [steps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
from miccai2021_cataract_semantic_segmentation_amd.utils.metrics import t_get_confusion_matrix

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=1e-4)
batches = [bench.synth_batch(8, 544, 960, 25, 1000 + 7919 * i, dev) for i in range(4)]
cm = torch.zeros((25, 25), dtype=torch.int32, device=dev)
fp = model.flat()


def eager(i):
    x, y = batches[i % 4]
    opt.zero_grad()
    out = model(x)
    loss = crit(*out, y)
    loss.backward()
    opt.step()
    t_get_confusion_matrix(out[1].detach(), y, cm)
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for i in range(n):
        h0 = time.perf_counter()
        fn(i)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, host / n * 1e3


for i in range(3):
    eager(i)
torch.cuda.synchronize()
w0, m0, v0 = fp.flat.clone(), opt._m.clone(), opt._v.clone()
b0 = [b.clone() for b in model.buffers()]
s0 = opt._steps
res = {}
res["eager_ms"], res["eager_host_ms"] = timed(eager, N)
w_e = fp.flat.clone()
with torch.no_grad():
    fp.flat.copy_(w0); opt._m.copy_(m0); opt._v.copy_(v0)
    for b, s in zip(model.buffers(), b0):
        b.copy_(s)
opt._steps = s0
t0 = time.perf_counter()
step = GraphedTrainStep(model, lambda o, l: crit(*o, l), opt, *batches[0], confusion=cm)
torch.cuda.synchronize()
res["capture_s"] = time.perf_counter() - t0
res["graph_ms"], res["graph_host_ms"] = timed(lambda i: step(*batches[i % 4]), N)
res["bit_identical_parameters_after_%d_steps" % N] = bool(torch.equal(fp.flat, w_e))
# second round each, alternating (box drift)
res["graph_ms_2"], res["graph_host_ms_2"] = timed(lambda i: step(*batches[i % 4]), N)
res["eager_ms_2"], res["eager_host_ms_2"] = timed(eager, N)
res["peak_hbm_GB"] = round(torch.cuda.max_memory_allocated() / 1e9, 1)
print(json.dumps(res, indent=1))
