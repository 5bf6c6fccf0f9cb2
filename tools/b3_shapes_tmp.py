import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch, torch.nn.functional as F
from miccai2021_cataract_semantic_segmentation_amd import ops, engine
from miccai2021_cataract_semantic_segmentation_amd.models import HRNetv2
from oracle.state import fill_state
g = np.load("/root/repo/tests/golden/hrnetv2_e3_tiny.npz")
spec = json.loads(str(g["spec"])); seed = int(g["seed"])
x = torch.from_numpy(g["x"])
ops.PRECISION = "fp32"
m = HRNetv2({}, 3); m.load_state_dict(fill_state(spec, seed)); m.cuda().train()
orig = ops.conv_fwd
rows = []
def hook(x, w, bias, Cout, kh, kw, stride=1, pad=0, dil=1, out=None, zero_to=0, stem4=False, groups=1):
    y = orig(x, w, bias, Cout, kh, kw, stride, pad, dil, out=out, zero_to=zero_to, stem4=stem4, groups=groups)
    if not stem4 and x.shape[-1] % 8 == 0 and w.dim() == 4:
        xc = x.contiguous()
        y64 = F.conv2d(xc.permute(0, 3, 1, 2).double(), w.double(), bias.double() if bias is not None else None, stride, pad, dil).permute(0, 2, 3, 1)
        yb = ops.conv_fwd_b3(tuple(x.shape), ops.split3(x), ops.split3_weight(w), bias, Cout, kh, kw, stride, pad, dil)
        sc = float(y64.abs().max())
        # error after removing the per-channel mean over positions (what BatchNorm sees), relative to the per-channel std
        def bn_err(t):
            d = (t.double() - y64).reshape(-1, Cout)
            std = y64.reshape(-1, Cout).std(0, unbiased=False) + 1e-30
            return float(((d - d.mean(0)) / std).abs().max()), float((d.mean(0) / std).abs().max())
        e32, eb = bn_err(y[..., :Cout]), bn_err(yb[..., :Cout])
        rows.append((tuple(x.shape), Cout, kh, stride, float((y[..., :Cout].double() - y64).abs().max()) / sc, float((yb[..., :Cout].double() - y64).abs().max()) / sc, e32, eb))
    return y
engine.ops.conv_fwd = hook
with torch.no_grad():
    m(x.cuda())
engine.ops.conv_fwd = orig
rows.sort(key=lambda r: -r[7][0] / (r[6][0] + 1e-12))
print("x shape, Cout, k, s | max err/scale fp32, b3 | BN-relative err (centred, mean-shift) fp32 | b3")
for r in rows[:25]:
    print(r[0], r[1], r[2], r[3], "| %.2e %.2e | %.2e %.2e | %.2e %.2e" % (r[4], r[5], r[6][0], r[6][1], r[7][0], r[7][1]))
print("layers:", len(rows), "median ratio centred", np.median([r[7][0] / (r[6][0] + 1e-12) for r in rows]))
