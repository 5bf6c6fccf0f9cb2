"""The ReLU mask of a residual block's output as bits (csrc/norm.hip: catseg_bn_apply_mask / catseg_bn_backward_mask; the BatchNorm + residual + ReLU
of the stage-1 bottlenecks, models/HRNetv2.py:68-106 of the reference): the bits are the signs of z, and the backward pass that reads them is
bit-identical to the one that reads z."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 12, 20, 256), (1, 5, 7, 512), (3, 33, 17, 256), (1, 1, 1, 1024), (2, 9, 11, 48), (1, 17, 30, 8), (2, 7, 5, 200)])
@pytest.mark.parametrize("acc", [False, True])
def test_mask_bits_and_backward_bit_identical(shape, acc):
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B + H * 7 + C)
    dev = torch.device("cuda")
    y = (torch.randn(B, H, W, C, generator=g) * 2 + 0.3).to(dev)
    res = torch.randn(B, H, W, C, generator=g).to(dev)
    dz = (torch.randn(B, H, W, C, generator=g) * 1e-3).to(dev)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats, scale = ops.bn_train_stats(y, gamma, 1e-5, 0.1, rm, rv)
    saved = ops.RELU_BITS
    try:
        ops.RELU_BITS = True
        assert ops.relu_bits_ok(y, res, True, None)
        z1 = ops.bn_apply(y, stats[:C], scale, beta, res, True, want_mask=True)
        ops.RELU_BITS = False
        z0 = ops.bn_apply(y, stats[:C], scale, beta, res, True, want_mask=True)
        torch.cuda.synchronize()
        assert getattr(z0, "_relu_mask", None) is None and getattr(z1, "_relu_mask", None) is not None
        assert torch.equal(z0, z1)
        # bit (e & 7) of byte e >> 3 = (flat element e is positive)
        m = z1._relu_mask.cpu().to(torch.int32)
        bits = ((m.unsqueeze(1) >> torch.arange(8, dtype=torch.int32).view(1, 8)) & 1).bool().reshape(-1)
        assert torch.equal(bits, (z1.cpu().reshape(-1) > 0))
        outs = []
        for zz in (z0, z1):
            dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
            dres = torch.full((B, H, W, C), 0.25, device=dev)
            dy = ops.bn_backward(dz, zz, y, stats, gamma, True, dgam, dbet, dres, acc)
            torch.cuda.synchronize()
            outs.append((dy.clone(), dgam.clone(), dbet.clone(), dres.clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    finally:
        ops.RELU_BITS = saved


def test_hrnet_step_takes_the_bits_and_reproduces_the_gradients():
    """OCRNet-HRNet step: the stage-1 bottlenecks' third BatchNorm carries its mask as bits; every gradient bit-identical to the z-reading route"""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    torch.manual_seed(8)
    net = OCRNet({"backbone": "hrnet48", "pretrained": False}, 3).cuda().train()
    x = torch.randn(2, 3, 64, 96, device="cuda")
    r = torch.randn(2, 25, 64, 96, device="cuda")
    saved = ops.RELU_BITS
    res = {}
    try:
        for mode in (True, False):
            ops.RELU_BITS = mode
            net.zero_grad()
            interm, final = net(x)
            (final * r).mean().backward()
            torch.cuda.synchronize()
            res[mode] = net.flat().grad.clone()
    finally:
        ops.RELU_BITS = saved
    assert torch.equal(res[True], res[False])


@pytest.mark.parametrize("shape", [(2, 12, 20, 96), (1, 9, 13, 192), (2, 5, 7, 384), (1, 130, 3, 48)])
@pytest.mark.parametrize("acc", [False, True])
def test_planes_route_mask_bits_and_backward_bit_identical(shape, acc):
    """the trunk's residual BatchNorm on the planes route (catseg_bn_apply_planes_mask / catseg_bn_backward_planes_mask): the mask bytes are the
    signs of z; dy planes, dgamma, dbeta and the residual gradient bit-identical to the z-reading route"""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B * 3 + H + C)
    dev = torch.device("cuda")
    y = (torch.randn(B, H, W, C, generator=g) * 2 + 0.3).to(dev)
    res = torch.randn(B, H, W, C, generator=g).to(dev)
    res._amax = ops.new_amax(dev)
    res._amax[0:1] = res.abs().max().reshape(1).view(torch.int32)
    dz = (torch.randn(B, H, W, C, generator=g) * 1e-3).to(dev)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats, scale = ops.bn_train_stats(y, gamma, 1e-5, 0.1, rm, rv)
    yrec = ops.new_amax(dev)
    yrec[0:1] = y.abs().max().reshape(1).view(torch.int32)
    bound = float(((gamma * stats[C:]).abs() * (y.abs().max() + stats[:C].abs()) + beta.abs()).max()) * 1.001
    saved = ops.RELU_BITS
    try:
        outs = []
        for mode in (True, False):
            ops.RELU_BITS = mode
            zrec = ops.new_amax(dev)
            zrec[2:3] = torch.tensor([bound], device=dev).view(torch.int32)          # CS_REC_BOUND, as bn_finalize(bound=...) leaves it
            z = ops.bn_apply(y, stats[:C], scale, beta, res, True, planes_rec=zrec, want_mask=True)
            assert (getattr(z, "_relu_mask", None) is not None) == mode
            if mode:
                m = z._relu_mask.cpu().to(torch.int32)
                bits = ((m.unsqueeze(1) >> torch.arange(8, dtype=torch.int32).view(1, 8)) & 1).bool().reshape(-1)
                assert torch.equal(bits, (z.cpu().reshape(-1) > 0))
            dgam, dbet = torch.empty(C, device=dev), torch.empty(C, device=dev)
            dres = torch.full((B, H, W, C), 0.25, device=dev)
            dyp = ops.bn_backward_planes(dz, z, y, stats, gamma, True, dgam, dbet, dres, acc, beta, yrec)
            torch.cuda.synchronize()
            outs.append((z.clone(), z._planes.buf.clone(), dyp.buf.clone(), dgam.clone(), dbet.clone(), dres.clone(), dyp.rec[:4].clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    finally:
        ops.RELU_BITS = saved
