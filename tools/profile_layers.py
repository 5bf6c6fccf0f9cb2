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
model = OCRNet(dict(bench.MODELS[name][0]), 3).to(dev).train()
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4}, "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=1e-4)
img, lbl = bench.synth_batch(8, 544, 960, 25, 1, dev)

# wrap the three conv entry points to remember shapes
shapes = []
_f, _d, _w = ops.conv_fwd, ops.conv_bwd_data, ops.conv_bwd_weight
def cf(x, w, b, Cout, kh, kw, *a, **k):
    shapes.append(("fwd", tuple(x.shape), Cout, kh, a[:3])); return _f(x, w, b, Cout, kh, kw, *a, **k)
def cd(dy, w, xs, kh, kw, *a, **k):
    shapes.append(("dgrad", tuple(xs), dy.shape[-1], kh, a[:3])); return _d(dy, w, xs, kh, kw, *a, **k)
def cw(x, dy, dw, db, kh, kw, *a, **k):
    shapes.append(("wgrad", tuple(x.shape), dy.shape[-1], kh, a[:3])); return _w(x, dy, dw, db, kh, kw, *a, **k)
import miccai2021_cataract_semantic_segmentation_amd.engine as E
def step():
    opt.zero_grad(); i, f = model(img); l = crit(i, f, lbl); l.backward(); opt.step()
for _ in range(2): step()
E.ops.conv_fwd, E.ops.conv_bwd_data, E.ops.conv_bwd_weight = cf, cd, cw
E.PARALLEL_BRANCHES = False   # sequential branches: the events around one launch then time that launch only
step(); torch.cuda.synchronize()
ops.PROFILE = []
step(); torch.cuda.synchronize()
prof = [q for q in ops.PROFILE if q[0].split('_')[0] in ('fwd', 'dgrad', 'wgrad') and q[0] not in ('dgrad_d3p', 'wgrad_d3p')]   # (the planes route's backward does not pass the wrapped entry points); ops.PROFILE = None
agg = collections.OrderedDict()
for (kind, fl, e0, e1), sh in zip(prof, shapes):
    assert kind.split('_')[0] == sh[0], (kind, sh)
    k = (kind,) + sh[1:]
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1)
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
tot = sum(v[2] for v in agg.values())
print("total igemm ms %.1f" % tot)
for k, (n, fl, ms) in rows[:120]:
    print("%-8s x%-3d in%-22s Cout %-5d k%d %-10s %8.2f ms %6.1f TF %5.1f%%" % (k[0], n, k[1], k[2], k[3], k[4], ms, fl / ms / 1e9, 100 * ms / tot))
