"""EncDec / UPerNet / ResNeXt (config 5 path) on the GPU."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("precision")]
T = torch.from_numpy


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _close(a, b, atol, rtol):
    a = a.detach().cpu().double().numpy()
    b = np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * np.abs(b).max(), "max abs err %g (scale %g)" % (err, np.abs(b).max())


@pytest.mark.parametrize("S,H,W", [(1, 9, 13), (2, 9, 13), (3, 17, 30), (6, 17, 30), (6, 6, 6), (6, 3, 4)])
def test_adaptive_avgpool(S, H, W):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(S * H)
    x = torch.randn(2, 24, H, W, generator=g, requires_grad=True)
    y = F.adaptive_avg_pool2d(x, S)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    yd = ops.adaptive_avgpool_fwd(xd, S)
    _close(yd.permute(0, 3, 1, 2), y.detach(), 1e-6, 1e-6)
    dx = torch.ones_like(xd)
    ops.adaptive_avgpool_bwd(gy.permute(0, 2, 3, 1).contiguous().cuda(), dx, S, True)
    _close(dx.permute(0, 3, 1, 2), x.grad + 1, 1e-6, 1e-6)


@pytest.mark.parametrize("case", [(1, 14, 18, 256, 256, 32, 1), (2, 9, 11, 512, 512, 32, 2), (1, 8, 8, 64, 128, 4, 1)])
def test_grouped_conv_forward(case):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, H, W, Ci, Co, G, s = case
    g = torch.Generator().manual_seed(Ci + G)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci // G, 3, 3, generator=g) * 0.1
    y = F.conv2d(x, w, None, s, 1, 1, G)
    yd = ops.conv_fwd(x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda().contiguous(memory_format=torch.channels_last), None,
                      Co, 3, 3, s, 1, 1, groups=G)
    _close(yd.permute(0, 3, 1, 2), y, 1e-4, 2e-4)


def test_encdec_resnet18_upernet_matches_reference_fixture(golden):
    _need_gpu()
    from oracle.state import fill_state
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    from miccai2021_cataract_semantic_segmentation_amd.losses import LossWrapper
    g = golden("encdec_r18_upernet_e1_tiny")
    spec = json.loads(str(g["spec"]))
    model = EncDec({"encoder": {"model": "ResNet18", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 1)
    assert [k for k, _ in spec] == list(model.state_dict().keys())
    model.load_state_dict(fill_state(spec, int(g["seed"])))
    model.cuda().eval()
    x, lbl = T(g["x"]).cuda(), T(g["lbl"]).cuda()
    model.get_features = False
    with torch.no_grad():
        _close(model(x), g["eval_final"], 0, 1e-3)
    model.train()
    model.get_features = True
    feat, y = model(x)
    _close(y, g["train_final"], 1e-3, 1e-3)
    _close(feat, g["train_feat"], 1e-3, 1e-3)
    crit = LossWrapper({"losses": {"LovaszSoftmax": 1}, "experiment": 1, "device": "cuda"})   # configs/UPN_rf_lvsz.json
    loss = crit(feat, y, lbl)
    assert abs(float(loss) - float(g["loss"])) < 2e-4
    loss.backward()
    names = json.loads(str(g["grad_names"]))
    P = dict(model.named_parameters())
    norms = np.array([float(P[k].grad.double().norm()) for k in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-2, atol=1e-6)
    # per-parameter gradients, calibrated against an fp64 evaluation of the oracle (tests/_calib.py)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _calib import calibrated_grad_check
    from oracle import losses as OL, upernet as OU
    calibrated_grad_check(model, spec, int(g["seed"]), lambda S_, x_: OU.encdec_forward(S_, x_, "ResNet18", train=True)[1],
                          lambda o, l: OL.lovasz_softmax(o, l), T(g["x"]), T(g["lbl"]), label="EncDec(ResNet18+UPerNet)")


def _resnext_oracle(S, x):
    from oracle.upernet import resnext_features_eval
    return resnext_features_eval(S, x)


def test_resnext101_upernet_inference_vs_oracle():
    """config-5 path (EncDec(ResNeXt101_32x8d + UPerNet), eval mode) against a CPU restatement"""
    _need_gpu()
    from oracle.state import fill_state, spec_of
    from oracle.upernet import upernet_forward
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    model = EncDec({"encoder": {"model": "ResNeXt101", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 3)
    S = fill_state(spec_of(model.state_dict()), 31)
    model.load_state_dict(S)
    model.cuda().eval()
    model.get_features = False
    x = torch.rand(1, 3, 96, 160, generator=torch.Generator().manual_seed(32))
    with torch.no_grad():
        y = model(x.cuda())
        ref = upernet_forward(S, _resnext_oracle(S, x), False)
    assert y.shape == (1, 25, 96, 160)
    _close(y, ref.numpy(), 0, 2e-3)
    bad = (y.argmax(1).cpu() != ref.argmax(1))
    top2 = ref.topk(2, dim=1).values
    assert bool(((top2[:, 0] - top2[:, 1])[bad] < 2e-3 * float(ref.abs().max())).all())


def test_eval_fused_conv_bn_matches_unfused():
    """inference fast path (BatchNorm folded into the conv, bias/residual/ReLU in the epilogue) == the 3-kernel path"""
    _need_gpu()
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd import engine
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    model = OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 3)
    model.load_state_dict(fill_state(spec_of(model.state_dict()), 77))
    model.cuda().eval()
    x = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        engine.FUSE_EVAL_BN = False
        a = model(x)[1].clone()
        engine.FUSE_EVAL_BN = True
        b = model(x)[1]
    _close(b, a.cpu().numpy(), 0, 1e-3)   # folding the BN scale into w changes the rounding order, ~2e-4 through 53 layers


def test_resnext50_upernet_trains():
    """EncDec(ResNeXt50_32x4d + UPerNet) as a TRAINING network (grouped-convolution backward): a few Adam steps reduce the loss
    and every parameter receives a finite gradient"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    from miccai2021_cataract_semantic_segmentation_amd.losses import LossWrapper
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    torch.manual_seed(3)
    model = EncDec({"encoder": {"model": "ResNeXt50", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 1).cuda().train()
    crit = LossWrapper({"losses": {"LovaszSoftmax": 1}, "experiment": 1, "device": "cuda"})
    opt = FusedAdam(model, lr=1e-3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 64, 96, generator=g).cuda()
    lbl = torch.randint(0, 8, (2, 8, 12), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2).cuda()
    losses = []
    for step in range(6):
        opt.zero_grad()
        feat, y = model(x)
        loss = crit(feat, y, lbl)
        loss.backward()
        if step == 0:
            grads = model.flat().grad
            assert bool(torch.isfinite(grads).all())
            w = model.enc_model.layer2[0].conv2.weight
            assert w.shape[1] * 32 == w.shape[0] and float(w.grad.abs().max()) > 0        # grouped 3x3: [O, I/32, 3, 3]
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0], losses


@pytest.mark.usefixtures("precision")
def test_config1_plumbing_shape_vs_oracle():
    """BASELINE config 1 ("FCN-ResNet18, 8-class (task 1), 2x3x256x256"): no such network exists in the reference;
    the mapping used here is its ResNet18 part, EncDec(ResNet18 + UPerNet), task 1 (K = 8), at exactly 2 x 3 x 256 x 256 with the
    LossWrapper / LovaszSoftmax loss of configs/UPN_rf_lvsz.json: logits, loss and calibrated gradients against the CPU oracle"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _calib import calibrated_grad_check
    from oracle import losses as OL, upernet as OU
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    from miccai2021_cataract_semantic_segmentation_amd.losses import LossWrapper
    model = EncDec({"encoder": {"model": "ResNet18", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 1)
    spec = spec_of(model.state_dict())
    S = fill_state(spec, 41)
    model.load_state_dict(S)
    model.cuda().train()
    g = torch.Generator().manual_seed(42)
    x = torch.rand(2, 3, 256, 256, generator=g)
    lbl = torch.randint(0, 8, (2, 16, 16), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2)
    crit = LossWrapper({"losses": {"LovaszSoftmax": 1}, "experiment": 1, "device": "cuda"})
    feat, y = model(x.cuda())
    assert y.shape == (2, 8, 256, 256)
    loss = crit(feat, y, lbl.cuda())
    loss.backward()
    with torch.no_grad():
        of, oy = OU.encdec_forward({k: v.clone() for k, v in S.items()}, x, "ResNet18", train=True)
    _close(y, oy.numpy(), 0, 1e-3)
    _close(feat, of.numpy(), 1e-4, 1e-3)
    assert abs(float(loss) - float(OL.lovasz_softmax(oy, lbl))) < 1e-4
    calibrated_grad_check(model, spec, 41, lambda S_, x_: OU.encdec_forward(S_, x_, "ResNet18", train=True)[1],
                          lambda o, l: OL.lovasz_softmax(o, l), x, lbl, label="config 1: EncDec(ResNet18+UPerNet) 2x3x256x256")
