"""csrc/stem7.hip: the torchvision ResNet stem's first convolution (nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False): the resnet50 / resnet101
backbones of models/OCR.py:58-61 and models/DeepLabv3Plus.py:32-38 of the reference) as a direct kernel that accumulates the 147 products of an
output in fp64 and rounds once -- forward (+ BatchNorm partials) against float64 F.conv2d for NCHW and NHWC-4 images, odd sizes and rows wider
than one 240-pixel segment; then through engine.conv_bn_act against the implicit-GEMM route."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from miccai2021_cataract_semantic_segmentation_amd import ops as o
    return o


@pytest.mark.parametrize("shape,layout", [((2, 64, 96), "nchw"), ((1, 37, 51), "nhwc4"), ((2, 9, 1100), "nchw"), ((3, 4, 4), "nhwc4"),
                                          ((1, 544, 960), "nchw")])
def test_forward_is_the_correctly_rounded_convolution_and_its_batchnorm_statistics(ops, shape, layout):
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H + W)
    x = torch.randn(B, 3, H, W, generator=g) * 1.7 + 0.3
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).contiguous(memory_format=torch.channels_last)
    y64 = F.conv2d(x.double(), w.double(), None, 2, 3)
    dev = torch.device("cuda")
    if layout == "nchw":
        xd = x.to(dev)
    else:
        xd = torch.zeros(B, H, W, 4)
        xd[..., :3] = x.permute(0, 2, 3, 1)
        xd = xd.to(dev)
    wd = w.to(dev)
    assert ops.stem7_ok(xd, wd, 7, 7, 2, 3, 1, 1)
    y, partials = ops.stem7_fwd(xd, wd, None, bn_stats=True)
    torch.cuda.synchronize()
    Ho, Wo = y64.shape[2:]
    assert tuple(y.shape) == (B, Ho, Wo, 64)
    yh = y.cpu().double().permute(0, 3, 1, 2)
    # products accumulated in fp64 (147 roundings of 2^-53), rounded once to fp32: half an ulp of every element (+ what the fp64 sum lost:
    # 147 * 2^-53 of the sum of |products|, which only matters for outputs that cancel to ~0)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, 2, 3)
    assert bool(((yh - y64).abs() <= y64.abs() * 2.0 ** -24 * 1.0001 + mag * 2.0 ** -44).all()), float(((yh - y64).abs() / (y64.abs() + 1e-30)).max())
    scale = float(y64.abs().max())
    gamma, rm, rv = torch.ones(64, device=dev), torch.zeros(64, device=dev), torch.ones(64, device=dev)
    stats, _ = ops.bn_finalize(partials, B * Ho * Wo, 64, gamma, 0.0, 0.1, rm, rv)
    y2 = y64.permute(1, 0, 2, 3).reshape(64, -1)
    assert float((stats[:64].cpu().double() - y2.mean(1)).abs().max()) <= 2e-5 * scale
    var = y2.var(1, unbiased=False)
    assert float((stats[64:].cpu().double() ** -2 - var).abs().max()) <= 1e-4 * float(var.max())
    # without statistics: the same bits
    y_b = ops.stem7_fwd(xd, wd, None, bn_stats=False)
    assert torch.equal(y_b, y)


def test_engine_takes_the_stem7_kernel_and_matches_the_implicit_gemm_route(ops):
    """OCRNet-R50's stem through engine.conv_bn_act: the direct kernel runs in the training forward (hbm:stem7 in the profile); logits and the
    stem's weight gradient against the implicit-GEMM route (ops.STEM7 = False), held to the step's own numerical sensitivity as in
    tests/test_stem3_gpu.py (the implicit-GEMM route re-run on an image perturbed by 1e-7 relative)"""
    import bench
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    torch.manual_seed(2)
    net = OCRNet(dict(bench.MODELS["ocrnet_r50"][0]), 3).cuda().train()
    x = torch.randn(2, 3, 96, 160, device="cuda")
    x2 = x * (1 + 1e-7 * torch.randn_like(x))
    r = torch.randn(2, 25, 96, 160, device="cuda")
    res = {}
    saved = ops.STEM7
    try:
        for tag, mode, inp in (("A", True, x), ("B", False, x), ("C", False, x2)):
            ops.STEM7 = mode
            net.zero_grad()
            ops.PROFILE = []
            interm, final = net(inp)
            (final * r).mean().backward()
            torch.cuda.synchronize()
            kinds = [p[0] for p in ops.PROFILE]
            ops.PROFILE = None
            assert (kinds.count("hbm:stem7") == 1) == mode, kinds[:10]
            res[tag] = (final.detach().clone().double(), net.backbone["conv1"].weight.grad.detach().clone().double())
    finally:
        ops.STEM7 = saved
        ops.PROFILE = None
    for k in (0, 1):
        scale = float(res["B"][k].abs().max())
        dab, dbc = float((res["A"][k] - res["B"][k]).abs().max()) / scale, float((res["B"][k] - res["C"][k]).abs().max()) / scale
        assert dab <= 4.0 * dbc + 1e-5, (k, dab, dbc)
    # eval mode keeps the fused inference path (folded BatchNorm): no stem7 launch
    net.eval()
    ops.PROFILE = []
    with torch.no_grad():
        net(x)
    torch.cuda.synchronize()
    kinds = [p[0] for p in ops.PROFILE]
    ops.PROFILE = None
    assert "hbm:stem7" not in kinds
