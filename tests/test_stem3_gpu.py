"""csrc/stem3.hip: the HRNet stem's first convolution (nn.Conv2d(3, 64, 3, stride 2, padding 1), models/HRNetv2.py:281-283 of the reference) as
direct fp32 kernels -- forward (+ BatchNorm partials) and backward-weight against float64 F.conv2d / autograd, for NCHW and NHWC-4 images,
odd sizes and rows wider than one 480-pixel segment."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from miccai2021_cataract_semantic_segmentation_amd import ops as o
    return o


@pytest.mark.parametrize("shape", [(2, 64, 96), (1, 37, 51), (2, 8, 2000), (3, 2, 2), (1, 544, 960)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc4"])
def test_forward_stats_and_backward_weight_vs_fp64(ops, shape, layout):
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.randn(B, 3, H, W, generator=g) * 1.7 + 0.3
    w = (torch.randn(64, 3, 3, 3, generator=g) * 0.2).contiguous(memory_format=torch.channels_last)
    xr, wr = x.double(), w.double().requires_grad_()
    y64 = F.conv2d(xr, wr, None, 2, 1)
    gy = torch.randn(y64.shape, generator=g) * 1e-3
    y64.backward(gy.double())
    dev = torch.device("cuda")
    if layout == "nchw":
        xd = x.to(dev)
    else:
        xd = torch.zeros(B, H, W, 4)
        xd[..., :3] = x.permute(0, 2, 3, 1)
        xd = xd.to(dev)
    wd = w.to(dev)
    assert ops.stem3_ok(xd, wd, 3, 3, 2, 1, 1, 1)
    y, partials = ops.stem3_fwd(xd, wd, None, bn_stats=True)
    torch.cuda.synchronize()
    Ho, Wo = y64.shape[2:]
    assert tuple(y.shape) == (B, Ho, Wo, 64)
    yh = y.cpu().double().permute(0, 3, 1, 2)
    scale = float(y64.abs().max())
    # the products accumulated in fp64, rounded once: the correctly rounded convolution (half an ulp of every element)
    ref = y64.detach()
    assert bool(((yh - ref).abs() <= ref.abs() * 2.0 ** -24 * 1.0001 + 1e-37).all()), float(((yh - ref).abs() / (ref.abs() + 1e-30)).max())
    # BatchNorm statistics from the kernel's partial rows
    gamma, rm, rv = torch.ones(64, device=dev), torch.zeros(64, device=dev), torch.ones(64, device=dev)
    stats, _ = ops.bn_finalize(partials, B * Ho * Wo, 64, gamma, 0.0, 0.1, rm, rv)
    y2 = y64.detach().permute(1, 0, 2, 3).reshape(64, -1)
    assert float((stats[:64].cpu().double() - y2.mean(1)).abs().max()) <= 2e-5 * scale
    if y2.shape[1] > 1:
        var = y2.var(1, unbiased=False)
        assert float((stats[64:].cpu().double() ** -2 - var).abs().max()) <= 1e-4 * float(var.max())
    # backward-weight
    dyd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    dw = torch.full((64, 3, 3, 3), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
    ops.stem3_bwd_weight(xd, dyd, dw)
    torch.cuda.synchronize()
    ref = wr.grad
    assert float((dw.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-12


def test_engine_takes_the_stem_kernels_and_matches_the_implicit_gemm_route(ops):
    """HRNet's stem through engine.conv_bn_act: the direct kernels run (hbm:stem3 in the profile); logits and the stem's weight gradient against
    the implicit-GEMM route (CATSEG_STEM3=0), held to the step's own numerical sensitivity: the implicit-GEMM route re-run on an image perturbed
    by 1e-7 relative (two fp32 evaluations of the first layer differ by an ulp; ~60 layers at 24 x 40 pixels amplify that to 1e-4 of the logits
    and 1e-2 of some gradients)"""
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    cfg = {"backbone": "hrnet18", "pretrained": False, "hrnet": {"width": 16, "stage1_width": 32, "modules": (1, 1, 1)}}
    torch.manual_seed(2)
    net = OCRNet(dict(cfg), 3).cuda().train()
    x = torch.randn(2, 3, 96, 160, device="cuda")
    x2 = x * (1 + 1e-7 * torch.randn_like(x))
    r = torch.randn(2, 25, 96, 160, device="cuda")
    res = {}
    saved = ops.STEM3
    try:
        for tag, mode, inp in (("A", True, x), ("B", False, x), ("C", False, x2)):
            ops.STEM3 = mode
            net.zero_grad()
            ops.PROFILE = []
            interm, final = net(inp)
            (final * r).mean().backward()
            torch.cuda.synchronize()
            kinds = [p[0] for p in ops.PROFILE]
            ops.PROFILE = None
            assert (kinds.count("hbm:stem3") == 2) == mode, kinds[:10]
            res[tag] = (final.detach().clone().double(), net.backbone.conv1.weight.grad.detach().clone().double())
    finally:
        ops.STEM3 = saved
        ops.PROFILE = None
    for k in (0, 1):
        scale = float(res["B"][k].abs().max())
        dab, dbc = float((res["A"][k] - res["B"][k]).abs().max()) / scale, float((res["B"][k] - res["C"][k]).abs().max()) / scale
        assert dab <= 4.0 * dbc + 1e-5, (k, dab, dbc)
