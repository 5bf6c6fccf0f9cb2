"""Per-layer igemm timing of one training step (HIP events around every conv launch)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from miccai2021_cataract_semantic_segmentation_amd import ops
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam

name = sys.argv[1] if len(sys.argv) > 1 else "ocrnet_hrnet48"
dev = torch.device("cuda")
if name.startswith("deeplab"):
    from miccai2021_cataract_semantic_segmentation_amd.models import DeepLabv3Plus
    from miccai2021_cataract_semantic_segmentation_amd.losses import CrossEntropyLoss
    model = DeepLabv3Plus(dict(bench.MODELS[name][0]), 2).to(dev).train()
    ce = CrossEntropyLoss(ignore_index=17)
    img, lbl = bench.synth_batch(8, 544, 960, 17, 1, dev)
    def loss_of(out):
        return ce(out, lbl)
else:
    model = OCRNet(dict(bench.MODELS[name][0]), 3).to(dev).train()
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    img, lbl = bench.synth_batch(8, 544, 960, 25, 1, dev)
    def loss_of(out):
        return crit(out[0], out[1], lbl)
opt = FusedAdam(model, lr=1e-4)

# wrap the three conv entry points: the PROFILE entries a call appends belong to its shape (routes that bypass the wrapped entry points --
# the planes / blocked-planes backward -- stay labelled by their kind only)
import miccai2021_cataract_semantic_segmentation_amd.engine as E
owner = {}
_f, _d, _w = ops.conv_fwd, ops.conv_bwd_data, ops.conv_bwd_weight
def _wrap(fn, label):
    def g(*a, **k):
        n0 = len(ops.PROFILE) if ops.PROFILE is not None else 0
        r = fn(*a, **k)
        if ops.PROFILE is not None:
            for i in range(n0, len(ops.PROFILE)):
                owner[i] = label(*a, **k)
        return r
    return g
cf = _wrap(_f, lambda x, w, b, Cout, kh, kw, *a, **k: (tuple(x.shape), Cout, kh, a[:3], "exact" if k.get("exact") else ""))
cd = _wrap(_d, lambda dy, w, xs, kh, kw, *a, **k: (tuple(xs), dy.shape[-1], kh, a[:3], ""))
cw = _wrap(_w, lambda x, dy, dw, db, kh, kw, *a, **k: (tuple(x.shape), dy.shape[-1], kh, a[:3], ""))
def step():
    opt.zero_grad(); l = loss_of(model(img)); l.backward(); opt.step()
for _ in range(2): step()
E.ops.conv_fwd, E.ops.conv_bwd_data, E.ops.conv_bwd_weight = cf, cd, cw
E.PARALLEL_BRANCHES = False   # sequential branches: the events around one launch then time that launch only
step(); torch.cuda.synchronize()
ops.PROFILE = []
step(); torch.cuda.synchronize()
prof = ops.PROFILE; ops.PROFILE = None
agg = collections.OrderedDict()
for i, (kind, fl, e0, e1) in enumerate(prof):
    k = (kind,) + owner.get(i, ("-", 0, 0, (), ""))
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1)
only = sys.argv[2] if len(sys.argv) > 2 else None      # e.g. "f32": kinds fwd / dgrad / wgrad (the fp32 family) only
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
if only == "f32":
    rows = [r for r in rows if r[0][0] in ("fwd", "dgrad", "wgrad")]
tot = sum(v[2] for _, v in rows)
print("total ms %.1f" % tot)
for k, (n, fl, ms) in rows[:150]:
    print("%-10s x%-3d in%-22s Cout %-5d k%d %-10s %-5s %8.3f ms %6.1f TF %5.1f%%" % (k[0], n, k[1], k[2], k[3], k[4], k[5], ms, fl / ms / 1e9 if ms else 0, 100 * ms / tot))
