"""The direct 3x3 kernels on producer-written fp16 x 2 planes (csrc/dconv3_pl.hip, csrc/planes.h) through the C ABI: against an fp64
F.conv2d (what the oracle's networks are built from: oracle.nets.conv), and BIT FOR BIT against the in-kernel-split kernel
catseg_dconv3_f16x2 they replace (same planes arithmetic, same product order, same K order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(48, 2, 19, 37), (48, 1, 8, 16), (48, 3, 5, 70), (64, 2, 9, 33), (96, 2, 19, 37), (96, 1, 4, 16), (192, 2, 7, 45), (192, 1, 2, 32),
          (384, 2, 5, 30), (384, 1, 3, 70), (48, 4, 40, 48)]


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _inputs(C, B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    chan = torch.exp(torch.randn(C, generator=g) * 1.5)                       # per-channel dynamic range e^+-4.5
    x = torch.randn(B, C, H, W, generator=g) * chan.view(1, C, 1, 1)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    return x, w


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("backward_data", [False, True])
def test_dconv3_pl_vs_fp64_and_bitwise_vs_in_kernel_split(shape, backward_data):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    C, B, H, W = shape
    dev = torch.device("cuda")
    x, w = _inputs(C, B, H, W, C + H)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last)
    if backward_data:
        ref = torch.nn.grad.conv2d_input(x.shape, w.double(), x.double(), 1, 1)     # dx for dy = x
    else:
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    try:
        wimg = ops.dconv3_weight_image(wd, backward_data=backward_data, h2=True)
        xp = ops.planes_from_f32(xd)
        rec_y = ops.new_amax(dev)
        y, (part, nr, _, cnt) = ops.dconv3_pl(xp, wimg, None, out=torch.full_like(xd, float("nan")), bn_stats=True, out_rec=rec_y)
        torch.cuda.synchronize()
        got = y.permute(0, 3, 1, 2).cpu().double()
        assert torch.isfinite(got).all()
        e = float((got - ref).abs().max()) / float(ref.abs().max())
        assert e <= 2e-5, e
        # max|y| in the output's record
        amax = rec_y.view(torch.float32)[::32].max()
        assert float(amax) == float(y.abs().max())
        # the in-kernel-split kernel with the same exponent (its amax record = the one the planes were made from): bit-identical
        y2 = ops.dconv3(xd, wimg, None, out=torch.full_like(xd, float("nan")), x_amax=xp.rec)
        assert torch.equal(y, y2), float((y - y2).abs().max())
        # per-wave BatchNorm partials -> statistics
        assert int(cnt.sum()) == B * H * W
        gamma = torch.ones(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        stats, scale = ops.bn_finalize((part, nr, 0, cnt), B * H * W, C, gamma, 1e-5, 0.1, rm, rv)
        yr = y.reshape(-1, C).double()
        mean, var = yr.mean(0), yr.var(0, unbiased=False)
        assert float((stats[:C].double() - mean).abs().max()) <= 2e-6 * float(yr.abs().max())
        assert float((stats[C:].double() - (var + 1e-5).rsqrt()).abs().max() / (var + 1e-5).rsqrt().max()) <= 2e-5
        # accumulate
        base = torch.randn_like(xd)
        y3 = ops.dconv3_pl(xp, wimg, None, out=base.clone(), accumulate=True)
        assert float((y3 - (base + y)).abs().max()) <= 1e-6 * float(y.abs().max())
    finally:
        ops.release_b3_cache()


@pytest.mark.parametrize("C", [48, 96, 192])
def test_planes_layout_and_exactness(C):
    """h + l reproduces x * 2^e to 2^-22 relative, laid out [plane][C / 8][pixel][8]; a record that overestimates max|x| is safe"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(2, 7, 13, C, generator=g) * torch.exp(torch.randn(C, generator=g) * 2)).to(dev)
    for over in (1.0, 6.0):
        rec = ops.new_amax(dev)
        rec[32:33] = (x.abs().max() * over).reshape(1).view(torch.int32)
        xp = ops.planes_from_f32(x, rec)
        e = int(xp.rec[1])
        P = 2 * 7 * 13
        pl = xp.buf.view(torch.float16).view(2, C // 8, P, 8).float()
        rebuilt = (pl[0] + pl[1]).permute(1, 0, 2).reshape(P, C) * 2.0 ** (-e)
        xf = x.reshape(P, C)
        assert float(((rebuilt - xf).abs() - xf.abs() * 2.0 ** -22).max()) <= 2.0 ** (-25 - e)      # (l is subnormal below 2^-17 max|x|)
        assert 2.0 ** 13 <= float(x.abs().max()) * over * 2.0 ** e < 2.0 ** 15


@pytest.mark.parametrize("shape", [(96, 2, 19, 37), (96, 1, 4, 16), (96, 3, 9, 50), (192, 2, 7, 45), (192, 1, 2, 32), (384, 2, 5, 30), (384, 1, 3, 70)])
def test_two_tiles_per_block_form_is_bit_identical(shape):
    """catseg_debug_set_dconv3_pl_pair: eight compute waves on two tiles that share the weight slots (one block per CU) -- the same per-wave
    arithmetic in the same order, so outputs, BatchNorm partials and the amax record equal the default form bit for bit; odd tile counts
    (a last unit with one tile) included"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    C, B, H, W = shape
    dev = torch.device("cuda")
    x, w = _inputs(C, B, H, W, 3 * C + W)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last)
    try:
        wimg = ops.dconv3_weight_image(wd, h2=True)
        xp = ops.planes_from_f32(xd)
        outs = []
        for mask in (0, 14):
            ops.lib.catseg_debug_set_dconv3_pl_pair(mask)
            rec = ops.new_amax(dev)
            y, (part, nr, _, cnt) = ops.dconv3_pl(xp, wimg, None, out=torch.full_like(xd, float("nan")), bn_stats=True, out_rec=rec)
            torch.cuda.synchronize()
            outs.append((y, part[:nr * 3 * C].clone(), cnt[:nr].clone(), float(rec.view(torch.float32)[::32].max())))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2]) and outs[0][3] == outs[1][3]
        live = outs[0][2] > 0                                      # (rows of empty waves are never written)
        pa, pb = outs[0][1].view(-1, 3 * C)[live], outs[1][1].view(-1, 3 * C)[live]
        assert torch.equal(pa, pb)
    finally:
        ops.lib.catseg_debug_set_dconv3_pl_pair(0)
        ops.release_b3_cache()


def test_planes_kernels_keep_their_blocks_per_cu():
    """the designs rest on co-residency: two blocks of the forward / backward-data kernel per CU (one in the two-tiles-per-block form), at least
    two of the backward-weight kernel (three at 96+ channels: a third image buffer would cost one, and did: 60 -> 92 us)"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    torch.cuda.init()
    for C in (48, 64, 96, 192, 384):
        assert ops.lib.catseg_debug_dconv3_pl_occupancy(C, 0) == 2, C
    for C in (96, 192, 384):
        assert ops.lib.catseg_debug_dconv3_pl_occupancy(C, 1) == 1, C
        assert ops.lib.catseg_debug_dwgrad3_pl_occupancy(C) >= 3, C
    assert ops.lib.catseg_debug_dwgrad3_pl_occupancy(48) >= 2
