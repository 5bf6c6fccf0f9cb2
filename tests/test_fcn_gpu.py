"""FCN-8s (reference models/FCN.py:7-61) on the GPU: the 2x2 max-pool and ConvTranspose2d kernels against ATen on the CPU, the network against the
fixture generated from the REAL reference (tests/golden/make_golden_fcn.py) and against the oracle, and the FCN manager's optimiser."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("precision")]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("shape", [(2, 8, 10, 12), (1, 20, 7, 9), (3, 4, 2, 2)])
def test_maxpool2x2_bit_exact(shape):
    """values and gradient routing identical to F.max_pool2d(x, 2), ties (ReLU zeros) and odd sizes included"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.relu(torch.randn(shape, generator=g)).requires_grad_()        # about half the entries tie at 0
    y = F.max_pool2d(x, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    yd, idx = ops.maxpool2_fwd(xd)
    assert torch.equal(yd.permute(0, 3, 1, 2).cpu(), y.detach())
    dx = torch.full((B, H, W, C), 7.0, device="cuda")
    ops.maxpool2_bwd(gy.permute(0, 2, 3, 1).contiguous().cuda(), idx, dx)
    assert torch.equal(dx.permute(0, 3, 1, 2).cpu(), x.grad)
    ops.maxpool2_bwd(gy.permute(0, 2, 3, 1).contiguous().cuda(), idx, dx, accumulate=True)
    assert torch.equal(dx.permute(0, 3, 1, 2).cpu(), 2 * x.grad)


@pytest.mark.parametrize("K,k,s,hw", [(17, 4, 2, (3, 4)), (25, 4, 2, (6, 8)), (17, 16, 8, (12, 16)), (8, 16, 8, (5, 3))])
def test_conv_transpose_matches_aten(K, k, s, hw):
    """nn.ConvTranspose2d (models/FCN.py:35-38: 4 / stride 2 and 16 / stride 8) forward and all three gradients"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    pad = (k - s + 1) // 2
    g = torch.Generator().manual_seed(K * k)
    x = torch.randn(2, K, *hw, generator=g).requires_grad_()
    w = (torch.randn(K, K, k, k, generator=g) * 0.1).requires_grad_()
    b = torch.randn(K, generator=g).requires_grad_()
    y = F.conv_transpose2d(x, w, b, s, pad)
    assert y.shape[-2:] == (hw[0] * s, hw[1] * s)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = ops.new_act(2, hw[0], hw[1], K, torch.device("cuda"), ld=32, zero=True)
    xd.copy_(x.detach().permute(0, 2, 3, 1))
    wd = w.detach().permute(0, 2, 3, 1).contiguous().cuda()          # physical [Cin][k][k][Cout]: what engine.FlatParams keeps
    yd, wp = ops.conv_transpose_fwd(xd, wd, b.detach().cuda(), K, k, s, pad)
    assert ops.ld_of(yd) == 32 and float(ops.widen(yd)[..., K:].abs().max()) == 0.0

    def close(a, ref, tol=2e-5):
        err = float((a.cpu().double() - ref.detach().double()).abs().max())
        assert err <= tol * float(ref.abs().max()), (err, float(ref.abs().max()))
    close(yd.permute(0, 3, 1, 2), y)
    gyd = ops.new_act(2, hw[0] * s, hw[1] * s, K, torch.device("cuda"), ld=32, zero=True)
    gyd.copy_(gy.permute(0, 2, 3, 1))
    dw, db = torch.empty_like(wd), torch.empty(K, device="cuda")
    dx = ops.new_act(2, hw[0], hw[1], K, torch.device("cuda"), ld=32, zero=True)
    ops.conv_transpose_bwd(gyd, xd, wp, dw, db, k, s, pad, dx, False)
    close(dx.permute(0, 3, 1, 2), x.grad)
    close(dw.permute(0, 3, 1, 2), w.grad, 1e-4)
    close(db, b.grad, 1e-4)
    assert float(ops.widen(dx)[..., K:].abs().max()) == 0.0
    ops.conv_transpose_bwd(gyd, xd, wp, dw, None, k, s, pad, dx, True)
    close(dx.permute(0, 3, 1, 2), 2 * x.grad)


def test_fcn_matches_reference_fixture_and_oracle(golden):
    _need_gpu()
    from make_golden_fcn import make_inputs, summarise, WIDTH
    from oracle import losses as OL, nets as ON
    from oracle.state import fill_state
    from miccai2021_cataract_semantic_segmentation_amd.models import FCN
    from miccai2021_cataract_semantic_segmentation_amd.losses import LovaszSoftmax
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    g = golden("fcn_w025_e2")
    spec = json.loads(str(g["spec"]))
    model = FCN({"width": WIDTH}, 2)
    assert [k for k, _ in spec] == list(model.state_dict().keys())
    assert [tuple(s) for _, s in spec] == [tuple(v.shape) for v in model.state_dict().values()]
    model.load_state_dict(fill_state(spec, int(g["seed"])))
    model.cuda().train()
    x, lbl = make_inputs()
    xd, ld = x.cuda(), lbl.cuda()
    scale = float(g["train_scale"])
    # the oracle's train step AND its fp64 forward from the same state: the 1e-3 bar is absolute, against fp64
    S = fill_state(spec, int(g["seed"]))
    with torch.no_grad():
        y64 = ON.fcn_forward({k: v.double() for k, v in S.items()}, x.double())
    for v in S.values():
        v.requires_grad_()
    yo = ON.fcn_forward(S, x)
    OL.lovasz_softmax(yo, lbl).backward()
    crit = LovaszSoftmax({"experiment": 2})
    opt = FusedAdam(model, lr=1e-3)
    losses = []
    for step in range(2):
        opt.zero_grad()
        y = model(xd)
        loss = crit(y, ld)
        loss.backward()
        if step == 0:
            yc = y.detach().cpu()
            s = summarise(yc)
            assert np.abs(s["sub"] - g["train_sub"]).max() <= 1e-3 * max(1.0, scale)
            assert np.abs(s["rows"] - g["train_rows"]).max() <= 1e-3 * max(1.0, scale)
            e64 = float((yc.double() - y64).abs().max())
            c64 = float((yo.detach().double() - y64).abs().max())
            print("FCN logits vs fp64: HIP %.3g, CPU fp32 %.3g (scale %.2f)" % (e64, c64, scale))
            assert e64 <= 1e-3
            names = json.loads(str(g["grad_names"]))
            P = dict(model.named_parameters())
            norms = np.array([float(P[k].grad.double().norm()) for k in names])
            np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-2, atol=1e-9)
            for k in names:          # every parameter's gradient against the oracle's, element by element
                ref = S[k].grad
                got = P[k].grad.cpu()
                err = float((got - ref).abs().max())
                assert err <= 3e-2 * float(ref.abs().max()) + 1e-9, (k, err, float(ref.abs().max()))
            for k in g.files:
                if k.startswith("g:") and k.endswith("[3]"):
                    ref = g[k]
                    assert np.abs(P[k[2:-3]].grad[3].cpu().numpy() - ref).max() <= 3e-2 * np.abs(ref).max(), k
        opt.step()
        losses.append(float(loss))
    assert abs(losses[0] - float(g["losses"][0])) < 1e-4
    assert abs(losses[1] - float(g["losses"][1])) < 5e-3 * float(g["losses"][1])
    # inference path (no tape) gives the same logits as the recorded forward of the same weights
    model.eval()
    with torch.no_grad():
        e1 = model(xd)
    model.train()
    y2 = model(xd)
    assert torch.equal(e1, y2.detach())


def test_fcn_rejects_sizes_the_reference_cannot_add():
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.models import FCN
    model = FCN({"width": 0.25}, 2).cuda().eval()
    with pytest.raises(ValueError, match="multiple of 32"):
        with torch.no_grad():
            model(torch.zeros(1, 3, 80, 96, device="cuda"))


@pytest.mark.parametrize("exp,hw", [(3, (128, 192)), (1, (256, 256))])
def test_fcn_full_width_against_oracle(exp, hw):
    """width 1 (64 .. 1024 channels) at 2 x 3 x 128 x 192 with 25 classes, and BASELINE config 1's shape (the reference's CPU-runnable case:
    8 classes, 2 x 3 x 256 x 256; SURVEY F8 maps it to FCN and to EncDec(ResNet18 + UPerNet)): logits within 1e-3 of the oracle's fp64
    forward, every parameter gradient against the oracle's fp32 train step"""
    _need_gpu()
    from oracle import losses as OL, nets as ON
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import FCN
    from miccai2021_cataract_semantic_segmentation_amd.losses import LovaszSoftmax
    model = FCN({"width": 1}, exp)
    K = model.num_classes
    spec = spec_of(model.state_dict())
    model.load_state_dict(fill_state(spec, 11))
    model.cuda().train()
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 3, *hw, generator=g)
    lbl = torch.randint(0, K + (exp != 1), (2, hw[0] // 16, hw[1] // 16), generator=g).repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
    S = fill_state(spec, 11)
    with torch.no_grad():
        y64 = ON.fcn_forward({k: v.double() for k, v in S.items()}, x.double())
    for v in S.values():
        v.requires_grad_()
    yo = ON.fcn_forward(S, x)
    lo = OL.lovasz_softmax(yo, lbl)
    lo.backward()
    y = model(x.cuda())
    loss = LovaszSoftmax({"experiment": exp})(y, lbl.cuda())
    loss.backward()
    e64 = float((y.detach().cpu().double() - y64).abs().max())
    c64 = float((yo.detach().double() - y64).abs().max())
    print("FCN width 1 logits vs fp64: HIP %.3g, CPU fp32 %.3g (scale %.2f)" % (e64, c64, float(y64.abs().max())))
    assert e64 <= max(1e-3, 1.5 * c64)
    assert abs(float(loss.detach()) - float(lo.detach())) < 1e-4
    for k, p in model.named_parameters():
        ref = S[k].grad
        err = float((p.grad.cpu() - ref).abs().max())
        assert err <= 3e-2 * float(ref.abs().max()) + 1e-9, (k, err, float(ref.abs().max()))
