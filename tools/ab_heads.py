"""within-run A/B of the head arithmetic (ops.HEADS: f16x2 = 2 fp16 planes / 3 products, bf16x3 = 3 bf16 planes / 6 products) on the
HRNet-W48 train step, plus the standalone times of the head layers"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
dev = torch.device("cuda")
def tm(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, W, Ci, Co, k, p) in [(8, 136, 240, 720, 512, 3, 1), (8, 136, 240, 1024, 512, 1, 0)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.02).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, W, Co, device=dev) * 1e-4
    y = torch.empty(B, H, W, Co, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    for heads in ("bf16x3", "f16x2", "bf16x3", "f16x2"):
        ops.HEADS = heads
        ops.release_b3_cache()
        ops.PROFILE = []
        for _ in range(6):
            ops.release_b3_cache()
            ops.conv_fwd(x, w, None, Co, k, k, 1, p, 1, out=y, train=True)
            ops.conv_bwd_weight(x, dy, dw, None, k, k, 1, p, 1)
            ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, 1, p, 1, out=dx)
        torch.cuda.synchronize()
        agg = {}
        for kind, fl, e0, e1 in ops.PROFILE[len(ops.PROFILE) // 3:]:
            a = agg.setdefault(kind, [0.0, 0]); a[0] += e0.elapsed_time(e1); a[1] += 1
        ops.PROFILE = None
        print("%dx%d %d->%d k%d %-7s " % (H, W, Ci, Co, k, heads) + "  ".join("%s %.2f ms" % (kk, v[0] / v[1] * (3 if kk == "split3" else 1)) for kk, v in sorted(agg.items())), flush=True)
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
for rnd in range(3):
    for heads in ("f16x2", "bf16x3"):
        ops.HEADS = heads
        print("round %d HEADS=%s: %.1f ms/step (loss %.7f)" % (rnd, heads, timeit(), float(step())), flush=True)
