"""Backward-weight of the trunk's 3x3 convolutions on producer-written planes (csrc/dwgrad3_pl.hip) through the C ABI: against the fp64
gradient of F.conv2d, and -- for the 48-channel configuration, whose tiling is unchanged -- BIT FOR BIT against catseg_dwgrad3_f16x2."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(48, 2, 19, 37), (48, 1, 4, 16), (48, 3, 5, 70), (96, 2, 19, 37), (96, 1, 2, 16), (192, 2, 7, 45), (384, 2, 5, 30), (384, 1, 3, 70),
          (48, 4, 40, 48), (96, 4, 34, 60)]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("shape", SHAPES)
def test_dwgrad3_pl_vs_fp64(shape):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    C, B, H, W = shape
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(C + W)
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(torch.randn(C, generator=g) * 1.5).view(1, C, 1, 1)
    dy = torch.randn(B, C, H, W, generator=g) * 1e-3 * torch.exp(torch.randn(C, generator=g)).view(1, C, 1, 1)
    ref = torch.nn.grad.conv2d_weight(x.double(), (C, C, 3, 3), dy.double(), 1, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    xp, dyp = ops.planes_from_f32(xd), ops.planes_from_f32(dyd)
    dw = torch.full((C, C, 3, 3), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
    ops.dwgrad3_pl(xp, dyp, dw)
    torch.cuda.synchronize()
    got = dw.cpu().double()
    assert torch.isfinite(got).all()
    e = float((got - ref).abs().max()) / float(ref.abs().max())
    assert e <= 2e-5, e
    dw2 = torch.full_like(dw, float("nan"))
    ops.dwgrad3_pl(xp, dyp, dw2)
    assert torch.equal(dw, dw2)                      # deterministic
    if C == 48:                                      # same tiles, same product and slab order as the in-kernel-split kernel
        xd._amax, dyd._amax = xp.rec, dyp.rec
        saved = ops.TRUNK
        ops.TRUNK = "f16x2"
        try:
            dw3 = torch.full_like(dw, float("nan"))
            ops.dwgrad3(xd, dyd, dw3)
        finally:
            ops.TRUNK = saved
        assert torch.equal(dw, dw3), float((dw - dw3).abs().max())
