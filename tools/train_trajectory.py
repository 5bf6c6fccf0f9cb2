"""Four Adam steps of the bench model (OCRNet-HRNetV2-W48, TwoScale Lovasz, lr 1e-4) at 2 x 3 x 544 x 960 on the HIP path and on the CPU oracle
from the same weights and the same two frames: the loss of every step, and after the last step the parameter update of both, compared.
A one-step gradient check cannot show an error that only the optimiser state or the BatchNorm running statistics carry from step to step;
this does.  Test infrastructure (the oracle is the checker); writes gpurun_out/train_trajectory.json.   python3 tools/train_trajectory.py [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from _fullres import block_labels  # noqa: E402
from oracle import losses as OL, nets as ON  # noqa: E402
from oracle.state import fill_state, spec_of  # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss  # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet  # noqa: E402
from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B, H, W, K, LR = 2, 544, 960, 25, 1e-4
torch.set_num_threads(int(os.environ.get("CATSEG_CPU_THREADS", "32")))
cfg = dict(bench.MODELS["ocrnet_hrnet48"][0])
model = OCRNet(dict(cfg), 3)
spec = spec_of(model.state_dict())
S0 = fill_state(spec, 41)
model.load_state_dict(S0)
model.cuda().train()
g = torch.Generator().manual_seed(9)
x = torch.rand(B, 3, H, W, generator=g)
lbl = block_labels(B, H, W, K, 10)

# ---- HIP path
crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                     "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
opt = FusedAdam(model, lr=LR)
xd, ld = x.cuda(), lbl.cuda()
hip_losses = []
for s in range(steps):
    opt.zero_grad()
    i, f = model(xd)
    loss = crit(i, f, ld)
    loss.backward()
    opt.step()
    hip_losses.append(float(loss.detach()))
torch.cuda.synchronize()
hip_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}

# ---- CPU oracle (fp32), torch.optim.Adam as the reference's managers build it (managers/BaseManager.py:441)
S = fill_state(spec, 41)
params = [v for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
for p in params:
    p.requires_grad_()
oopt = torch.optim.Adam(params, lr=LR)
cpu_losses, t0 = [], time.time()
for s in range(steps):
    oopt.zero_grad()
    oi, of = ON.ocrnet_hrnet_forward(S, x, train=True)
    ol = OL.two_scale_lovasz(oi, of, lbl, 0.4, 1.0)
    ol.backward()
    oopt.step()
    cpu_losses.append(float(ol.detach()))
    print("oracle step %d: loss %.6f (HIP %.6f)  %.0f s" % (s, cpu_losses[-1], hip_losses[s], time.time() - t0), flush=True)

# ---- the same trajectory in fp64 (the yardstick: how far does the fp32 CPU reference itself drift from exact arithmetic?)
S64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, 41).items()}
p64 = [v for k, v in S64.items() if v.dtype.is_floating_point and "running" not in k]
for p in p64:
    p.requires_grad_()
opt64 = torch.optim.Adam(p64, lr=LR)
f64_losses = []
for s in range(steps):
    opt64.zero_grad()
    oi, of = ON.ocrnet_hrnet_forward(S64, x.double(), train=True)
    ol = OL.two_scale_lovasz(oi, of, lbl, 0.4, 1.0)
    ol.backward()
    opt64.step()
    f64_losses.append(float(ol.detach()))
    print("fp64 step %d: loss %.6f  %.0f s" % (s, f64_losses[-1], time.time() - t0), flush=True)


def update_err(sd_a, sd_b):
    """relative distance of two parameter updates theta_T - theta_0 over all float tensors, and the share of elements whose update has the
    opposite sign"""
    num = den = 0.0
    flips = n = 0
    for k, v0 in S0.items():
        if not v0.dtype.is_floating_point or "running" in k:
            continue
        a, b = sd_a[k].detach().double() - v0.double(), sd_b[k].detach().double() - v0.double()
        num += float((a - b).norm()) ** 2
        den += float(b.norm()) ** 2
        flips += int(((a * b) < 0).sum())
        n += a.numel()
    return (num / den) ** 0.5, flips / n


# ---- compare: losses per step; parameter updates (theta_T - theta_0) tensor by tensor; BatchNorm running statistics
rows, worst = [], {"rel_update_err": 0.0, "cos": 1.0}
tot_num = tot_den = 0.0
for k, v0 in S0.items():
    if not v0.dtype.is_floating_point:
        continue
    a, b = hip_sd[k].double() - v0.double(), S[k].detach().double() - v0.double()
    den = float(b.norm())
    if den == 0.0:
        continue
    err = float((a - b).norm()) / den
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
    tot_num += float((a - b).norm()) ** 2
    tot_den += den ** 2
    rows.append((k, err, cos))
rows.sort(key=lambda r: -r[1])
res = {"config": "OCRNet-HRNetV2-W48, task 3, %d x 3 x %d x %d, TwoScale Lovasz, Adam lr %g, %d steps" % (B, H, W, LR, steps),
       "loss_hip": hip_losses, "loss_cpu_fp32": cpu_losses,
       "max_abs_loss_diff": max(abs(a - b) for a, b in zip(hip_losses, cpu_losses)),
       "update_rel_err_all_tensors": (tot_num / tot_den) ** 0.5,
       "loss_cpu_fp64": f64_losses,
       "abs_loss_diff_vs_fp64": {"hip": [abs(a - b) for a, b in zip(hip_losses, f64_losses)],
                                 "cpu_fp32": [abs(a - b) for a, b in zip(cpu_losses, f64_losses)]},
       "update_vs_fp64": {"hip": dict(zip(("rel_err", "opposite_sign_share"), update_err(hip_sd, S64))),
                          "cpu_fp32": dict(zip(("rel_err", "opposite_sign_share"), update_err(S, S64)))},
       "update_rel_err_worst_tensors": [{"key": k, "rel_err": e, "cosine": c} for k, e, c in rows[:8]],
       "update_cosine_min": min(c for _, _, c in rows),
       "n_tensors": len(rows)}
out = os.path.join(ROOT, "gpurun_out", "train_trajectory.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "update_rel_err_worst_tensors"}))
for r in res["update_rel_err_worst_tensors"][:5]:
    print(r)
