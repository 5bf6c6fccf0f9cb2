import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from miccai2021_cataract_semantic_segmentation_amd import ops, _lib
from miccai2021_cataract_semantic_segmentation_amd.models import HRNetv2
from miccai2021_cataract_semantic_segmentation_amd.losses import CrossEntropyLoss
from oracle.state import fill_state
from oracle import hrnet as OH, losses as OL
g = np.load("/root/repo/tests/golden/hrnetv2_e3_tiny.npz")
spec = json.loads(str(g["spec"])); seed = int(g["seed"])
x, lbl = torch.from_numpy(g["x"]), torch.from_numpy(g["lbl"])
grads = {}
for dt in (torch.float32, torch.float64):
    S = {k: (v.to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, seed).items()}
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params: S[k].requires_grad_()
    OL.cross_entropy(OH.hrnetv2_forward(S, x.to(dt), train=True), lbl, 3).backward()
    grads[dt] = {k: S[k].grad.double() for k in params}
calls = {"fwd": 0, "dgrad": 0}
of, od = _lib.lib.catseg_conv2d_fwd_bf16x3, _lib.lib.catseg_conv2d_bwd_data_bf16x3
def run(prec, opsel, nocache=False, tile=0):
    ops.PRECISION = prec; ops.B3_OPS = opsel
    ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = (1, 64, 32, 1)
    _lib.lib.catseg_debug_set_b3_tile(tile)
    saved = ops._split3_cached
    if nocache: ops._split3_cached = ops.split3
    m = HRNetv2({}, 3); m.load_state_dict(fill_state(spec, seed)); m.cuda().train()
    ops.PROFILE = []
    y = m(x.cuda()); loss = CrossEntropyLoss(ignore_index=25)(y, lbl.cuda()); loss.backward()
    torch.cuda.synchronize()
    n_f = sum(1 for p in ops.PROFILE if p[0] == "fwd"); n_d = sum(1 for p in ops.PROFILE if p[0] == "dgrad"); ops.PROFILE = None
    ops._split3_cached = saved
    P = dict(m.named_parameters()); r = []
    for k, g64 in grads[torch.float64].items():
        n = float(g64.norm())
        if n < 1e-7: continue
        e32 = float((grads[torch.float32][k] - g64).norm()); eh = float((P[k].grad.cpu().double() - g64).norm())
        r.append(eh / (e32 + 1e-4 * n))
    r = np.array(r)
    print(prec, opsel, "nocache" if nocache else "", "tile", tile, "median %.2f p95 %.2f max %.2f loss %.7f" % (np.median(r), np.percentile(r, 95), r.max(), float(loss)), flush=True)
run("fp32", ("fwd", "dgrad"))
run("bf16x3", ("fwd",))
run("bf16x3", ("fwd",), nocache=True)
run("bf16x3", ("dgrad",))
for t in (1, 2, 6, 9):
    run("bf16x3", ("fwd",), tile=t)
