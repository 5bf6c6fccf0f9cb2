"""what each kernel population costs ON THE CRITICAL PATH of the overlapped HRNet-W48 step: the step is timed with one population's
launches skipped (backward only: the forward, the loss and therefore the work of every other kernel stay what they are; the
skipped kernels' outputs are garbage, lr = 0).  Kernel-time sums over-count what overlaps across streams; this does not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
lib = ops.lib
dev = torch.device("cuda")
torch.manual_seed(0)
model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=0.0)
img, lbl = bench.synth_batch(8, 544, 960, 25, 1000, dev)
def step():
    opt.zero_grad(); i, f = model(img); loss = crit(i, f, lbl); loss.backward(); opt.step(); return loss
def timeit(n=6):
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
real = {}
def knock(names, pred=None):
    for n in names:
        f = getattr(lib, n)
        real[n] = f
        if pred is None or n != "catseg_dconv3":
            setattr(lib, n, lambda *a: 0)
        else:
            setattr(lib, n, (lambda f: (lambda *a: 0 if pred(a) else f(*a)))(f))
def restore():
    for n, f in real.items():
        setattr(lib, n, f)
    real.clear()
is_dgrad = lambda a: a[7] is None and a[11] is None          # catseg_dconv3(..., bias, ..., bn_part, ...): backward-data calls have neither
SETS = [
    ("nothing", [], None),
    ("BatchNorm backward (all)", ["catseg_bn_backward", "catseg_bn_backward_pre"], None),
    ("direct backward-weight (trunk)", ["catseg_dwgrad3"], None),
    ("direct backward-data (trunk)", ["catseg_dconv3", "catseg_dconv3_bnbwd"], is_dgrad),
    ("fp32 backward-weight", ["catseg_conv2d_bwd_weight"], None),
    ("fp32 backward-data", ["catseg_conv2d_bwd_data"], None),
    ("bf16x3 backward-weight (heads)", ["catseg_conv2d_bwd_weight_bf16x3"], None),
    ("bf16x3 backward-data (heads)", ["catseg_conv2d_bwd_data_bf16x3", "catseg_conv2d_bwd_data_bf16x3_blocked"], None),
    ("bilinear backward", ["catseg_bilinear_bwd"], None),
    ("nothing", [], None),
]
base = None
for name, fns, pred in SETS:
    fns = [f for f in fns if hasattr(lib, f)]
    knock(fns, pred)
    t = timeit()
    restore()
    if base is None:
        base = t
    print("without %-34s %7.1f ms/step  (%+6.1f)" % (name, t, t - base), flush=True)
