"""The BENCH workload at its own widths: OCRNet(backbone='hrnet48') built exactly as bench.py builds it, against the
CPU oracle (logits, loss, fp64-calibrated per-parameter gradients), and every convolution shape of that network
(HRNetV2-W48 trunk 48/96/192/384 + the 720-channel OCR heads) against F.conv2d with the tile forms the planner
picks for those layers AT THE BENCH SIZE (bs 8 @ 544x960) forced onto a CPU-checkable input.

("the bench workload is not the tested workload" -- the adjoint / linearity properties of
tests/test_fullsize_gpu.py would pass a consistently wrong tap order; these tests would not.)"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("precision")]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _close(a, b, atol, rtol):
    a = a.detach().cpu().double().numpy()
    b = np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * np.abs(b).max(), "max abs err %g (scale %g)" % (err, np.abs(b).max())


# (one CPU-checkable size here: the same model at the full 544 x 960 under the production plan is tests/test_fullres_gpu.py; a second
#  small size, 96 x 160, cost 160 s of the GPU suite's budget for the same code paths)
@pytest.mark.parametrize("size", [(64, 96)], ids=["64x96"])
def test_ocrnet_hrnet48_bench_model_vs_oracle(size):
    _need_gpu()
    import bench
    from oracle import nets as ON, losses as OL
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    H, W = size
    model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3)          # the bench's constructor call
    assert sum(p.numel() for p in model.backbone.parameters()) > 60e6   # W48 trunk (48/96/192/384), not a toy width
    spec = spec_of(model.state_dict())
    S = fill_state(spec, 31)
    model.load_state_dict(S)
    model.cuda().train()
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, H, W, generator=gen)
    lbl = torch.randint(0, 26, (2, H // 8, W // 8), generator=gen).repeat_interleave(8, 1).repeat_interleave(8, 2)
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    interm, final = model(x.cuda())
    loss = crit(interm, final, lbl.cuda())
    loss.backward()
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        S[k].requires_grad_()
    oi, of = ON.ocrnet_hrnet_forward(S, x, train=True)
    ol = OL.two_scale_lovasz(oi, of, lbl, 0.4, 1.0)
    ol.backward()
    # logits: 1e-3 of the logit scale; loss 1e-4
    _close(final, of.detach().numpy(), 0, 1e-3)
    _close(interm, oi.detach().numpy(), 0, 1e-3)
    assert abs(float(loss) - float(ol)) < 1e-4, (float(loss), float(ol))
    # BatchNorm running statistics after one training forward (momentum 0.01 in the trunk, 0.1 in the heads)
    sd = model.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.stage4.2.branches.3.3.bn2.running_var", "conv_high_map.1.running_var"):
        _close(sd[k], S[k].detach().numpy(), 1e-5, 1e-4)
    # logits and gradients calibrated against an fp64 evaluation of the oracle (tests/_calib.py)
    S64 = {k: (v.detach().double() if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, 31).items()}
    with torch.no_grad():
        f64 = ON.ocrnet_hrnet_forward(S64, x.double(), train=True)[1]
    e_cpu = float((of.detach().double() - f64).abs().max())
    e_hip = float((final.detach().cpu().double() - f64).abs().max())
    print("W48 %dx%d: |logit - fp64| cpu32 %.3g hip %.3g" % (H, W, e_cpu, e_hip))
    assert e_hip <= 3 * e_cpu + 1e-5, (e_hip, e_cpu)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _calib import calibrated_grad_check
    calibrated_grad_check(model, spec, 31, lambda S_, x_: ON.ocrnet_hrnet_forward(S_, x_, train=True),
                          lambda o, l: OL.two_scale_lovasz(o[0], o[1], l, 0.4, 1.0), x, lbl, label="OCRNet-HRNet-W48 %dx%d" % (H, W))


# ----------------------------------------------------------------------------------------------------------------------
# every convolution shape of OCRNet-HRNetV2-W48 at the bench size: (name, H, W at bs 8 / 3x544x960, Cin, Cout, k, stride)
W48_LAYERS = [
    ("stem2 64-64 s2", 272, 480, 64, 64, 3, 2), ("layer1 1x1 64-64", 136, 240, 64, 64, 1, 1), ("layer1 3x3 64-64", 136, 240, 64, 64, 3, 1),
    ("layer1 1x1 64-256", 136, 240, 64, 256, 1, 1), ("layer1 1x1 256-64", 136, 240, 256, 64, 1, 1),
    ("transition 256-48", 136, 240, 256, 48, 3, 1), ("transition 256-96 s2", 136, 240, 256, 96, 3, 2),
    ("transition 96-192 s2", 68, 120, 96, 192, 3, 2), ("transition 192-384 s2", 34, 60, 192, 384, 3, 2),
    ("branch 48", 136, 240, 48, 48, 3, 1), ("branch 96", 68, 120, 96, 96, 3, 1), ("branch 192", 34, 60, 192, 192, 3, 1),
    ("branch 384", 17, 30, 384, 384, 3, 1),
    ("fuse 1x1 96-48", 68, 120, 96, 48, 1, 1), ("fuse 1x1 192-48", 34, 60, 192, 48, 1, 1), ("fuse 1x1 384-48", 17, 30, 384, 48, 1, 1),
    ("fuse 1x1 192-96", 34, 60, 192, 96, 1, 1), ("fuse 1x1 384-96", 17, 30, 384, 96, 1, 1), ("fuse 1x1 384-192", 17, 30, 384, 192, 1, 1),
    ("fuse s2 48-96", 136, 240, 48, 96, 3, 2), ("fuse s2 48-48", 136, 240, 48, 48, 3, 2), ("fuse s2 48-192", 68, 120, 48, 192, 3, 2),
    ("fuse s2 48-384", 34, 60, 48, 384, 3, 2), ("fuse s2 96-96", 68, 120, 96, 96, 3, 2), ("fuse s2 96-384", 34, 60, 96, 384, 3, 2),
    ("head 3x3 720-512", 136, 240, 720, 512, 3, 1), ("classifier 512-25", 136, 240, 512, 25, 1, 1),
    ("ocr 1x1 512-256", 136, 240, 512, 256, 1, 1), ("ocr 1x1 256-256", 136, 240, 256, 256, 1, 1), ("ocr 1x1 256-512", 136, 240, 256, 512, 1, 1),
    ("ocr 1x1 1024-512", 136, 240, 1024, 512, 1, 1),
]


def _plan(lib, _lib, B, H, W, Ci, Co, k, s):
    from miccai2021_cataract_semantic_segmentation_amd.ops import conv_out_size
    p = k // 2
    d = _lib.ConvDesc(B, H, W, Ci, conv_out_size(H, k, s, p, 1), conv_out_size(W, k, s, p, 1), Co, k, k, s, p, 1,
                      Ci, (Co + 3) // 4 * 4, 0, 1)
    res = []
    for op in range(3):
        out = (ctypes.c_int * 5)()
        _lib.check(lib.catseg_debug_plan_conv(ctypes.byref(d), op, out))
        res.append(tuple(out))
    return res


@pytest.mark.parametrize("layer", W48_LAYERS, ids=[l[0] for l in W48_LAYERS])
def test_w48_conv_shape_with_bench_tiles_vs_torch(layer):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import _lib, ops
    lib = _lib.lib
    name, Hb, Wb, Ci, Co, k, s = layer
    p = k // 2
    plans = _plan(lib, _lib, 8, Hb, Wb, Ci, Co, k, s)
    # CPU-checkable size: >= 1000 output pixels, odd extents (partial tiles, every border); the direct backward-weight
    # kernel (48->48, 96->96 branches) wants >= 1024 strips of 16 pixels
    B, H, W = (2, 21, 27) if s == 1 else (2, 43, 55)
    if plans[2][4]:
        B, H, W = 4, 64, 70
    g = torch.Generator().manual_seed(Ci * 7 + Co + k + s)
    x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5).requires_grad_()
    b = torch.randn(Co, generator=g, requires_grad=True)
    y = F.conv2d(x, w, b, s, p, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.detach().cuda().contiguous(memory_format=torch.channels_last)
    gyd = ops.new_act(B, y.shape[2], y.shape[3], Co, xd.device, zero=True)
    gyd.copy_(gy.permute(0, 2, 3, 1))

    def run(forced):
        out = {}
        try:
            for op, key in ((0, "y"), (1, "dx"), (2, "dw")):
                if forced:
                    mi, ni, form, splits, _ = plans[op]
                    lib.catseg_debug_set_tile(mi + 16 * form, ni)
                    lib.catseg_debug_set_splits(min(splits, 12) if op == 2 else 0)
                if op == 0:
                    out["y"] = ops.conv_fwd(xd, wd, b.detach().cuda(), Co, k, k, s, p, 1)
                elif op == 1:
                    out["dx"] = ops.conv_bwd_data(gyd, wd, tuple(xd.shape), k, k, s, p, 1)
                else:
                    dw, db = torch.empty_like(wd), torch.empty(Co, device="cuda")
                    ops.conv_bwd_weight(xd, gyd, dw, db, k, k, s, p, 1)
                    out["dw"], out["db"] = dw, db
        finally:
            lib.catseg_debug_set_tile(0, 0)
            lib.catseg_debug_set_splits(0)
        return out

    for forced in (True, False):
        o = run(forced)
        _close(o["y"].permute(0, 3, 1, 2), y.detach().numpy(), 2e-5, 2e-5)
        _close(o["dx"].permute(0, 3, 1, 2), x.grad.numpy(), 2e-5, 2e-5)
        _close(o["dw"], w.grad.numpy(), 1e-4, 1e-4)
        _close(o["db"], b.grad.numpy(), 1e-4, 1e-4)


def test_second_forward_before_the_first_backward_keeps_the_amax_records(precision):
    """the f16x2 trunk kernels scale their operands by amax records that the producing kernels fill; the records of a
    forward pass must survive a SECOND forward pass (another micro-batch, another network) that runs before the first one's backward:
    gradients bit-identical to the plain forward -> backward order"""
    _need_gpu()
    if precision != "bf16x3":
        pytest.skip("the split-precision arithmetics only")
    import bench
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd import ops
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    assert ops.TRUNK == "f16x2"
    model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3)
    model.load_state_dict(fill_state(spec_of(model.state_dict()), 31))
    model.cuda().train()
    gen = torch.Generator().manual_seed(6)
    x1, x2 = torch.rand(2, 3, 64, 96, generator=gen).cuda(), (40.0 * torch.rand(2, 3, 64, 96, generator=gen)).cuda()
    lbl = torch.randint(0, 26, (2, 8, 12), generator=gen).repeat_interleave(8, 1).repeat_interleave(8, 2).cuda()
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    ops.PROFILE = []
    model.zero_grad()
    i1, f1 = model(x1)
    crit(i1, f1, lbl).backward()
    kinds = {k for k, *_ in ops.PROFILE}
    ops.PROFILE = None
    assert ({"fwd_d3p", "dgrad_d3p", "wgrad_d3p"} if ops.PLANES else {"fwd_d3h", "dgrad_d3h", "wgrad_d3h"}) <= kinds, kinds
    ref = model.flat().grad.clone()
    model.zero_grad()
    i1, f1 = model(x1)
    model(x2)                                  # a second recorded forward (40x larger activations) before the first backward
    crit(i1, f1, lbl).backward()
    assert torch.equal(model.flat().grad, ref)
