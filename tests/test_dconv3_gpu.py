"""Direct split-precision 3x3 convolution of the HRNet trunk widths (csrc/dconv3_b3.hip, reference call sites models/HRNetv2.py:22-25,41-44)
against an fp64 F.conv2d: forward (+bias, +BatchNorm partial statistics), backward-data (= the same kernel on dy with the mirrored
weight image, write and accumulate), ragged tiles, tensors living inside wider concat buffers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from miccai2021_cataract_semantic_segmentation_amd import ops as o
    return o


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.cpu().permute(0, 3, 1, 2)


def close(a, b, rtol):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-12
    assert err <= rtol * scale, "max abs err %g (ref scale %g)" % (err, scale)


# B, H, W, C: tile-aligned, ragged in both directions, smaller than one tile, one pixel row / column
CASES = [(2, 16, 32, 48), (1, 19, 37, 48), (2, 5, 7, 48), (1, 1, 50, 48), (3, 33, 1, 48),
         (2, 8, 64, 96), (1, 19, 37, 96), (2, 3, 5, 96), (1, 1, 70, 96), (2, 41, 2, 96),
         (1, 24, 40, 192), (2, 7, 9, 192), (1, 17, 30, 384), (1, 5, 3, 384), (2, 16, 32, 64), (1, 19, 37, 64)]


@pytest.fixture(params=[0, 1], ids=["uniform", "specialised"])
def spec(request):
    """both variants of the direct kernel: uniform waves, and 4 compute + 4 helper waves per block"""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    lib.catseg_debug_set_dconv3_spec(request.param)
    yield request.param
    lib.catseg_debug_set_dconv3_spec(-1)


@pytest.mark.parametrize("case", CASES)
def test_dconv3_forward_backward_data_vs_fp64(ops, case, spec):
    """error of fp32 size (<= 2e-5 of the output scale, as for the other bf16x3 kernels) on inputs with a wide dynamic range"""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    B, H, W, C = case
    if not lib.catseg_dconv3_supported(C):
        pytest.skip("width not built")
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(2 * torch.randn(1, C, 1, 1, generator=g))
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    b = torch.randn(C, generator=g)
    xr = x.double().requires_grad_()
    y64 = F.conv2d(xr, w.double(), b.double(), 1, 1, 1)
    gy = torch.randn(y64.shape, generator=g)
    y64.backward(gy.double())
    xd, wd = nhwc(x), w.cuda().contiguous(memory_format=torch.channels_last)
    wimg = ops.dconv3_weight_image(wd)
    y, part = ops.dconv3(xd, wimg, b.cuda(), bn_stats=True)
    close(nchw(y), y64.detach(), 2e-5)
    # BatchNorm statistics from the epilogue's per-tile partials
    gamma, rm, rv = torch.ones(C).cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda()
    stats, _ = ops.bn_finalize(part, B * H * W, C, gamma, 1e-5, 0.1, rm, rv)
    y2 = y64.detach().permute(1, 0, 2, 3).reshape(C, -1)
    mean, var = y2.mean(1), y2.var(1, unbiased=False)
    assert float((stats[:C].cpu().double() - mean).abs().max()) <= 2e-5 * float(y2.abs().max())
    assert float((stats[C:].cpu().double() * (var + 1e-5).sqrt() - 1).abs().max()) <= 1e-4
    # backward-data: write, then accumulate on top
    wimg_t = ops.dconv3_weight_image(wd, backward_data=True)
    gyd = nhwc(gy)
    dx = ops.dconv3(gyd, wimg_t)
    close(nchw(dx), xr.grad, 2e-5)
    dx2 = ops.dconv3(gyd, wimg_t, out=dx.clone(), accumulate=True)
    close(nchw(dx2), 2 * xr.grad, 2e-5)
    ops.release_b3_cache()


BQ_CASES = [(2, 16, 32, 48), (1, 19, 37, 48), (2, 5, 7, 48), (2, 8, 64, 96), (1, 19, 37, 96), (1, 24, 40, 192), (2, 7, 9, 192),
            (1, 17, 30, 384), (2, 16, 32, 64), (1, 19, 37, 64)]


@pytest.mark.parametrize("h2", [False, True], ids=["bf16x3", "f16x2"])
@pytest.mark.parametrize("case", BQ_CASES)
def test_backward_data_with_fused_bn_backward_pass(ops, case, spec, h2):
    """out = relu(bn1(q)); y = conv2(out) (models/HRNetv2.py:36-47): the backward-data launch of conv2 masks its result with
    relu(bn1(q)) > 0 and leaves the per-tile sums of the first pass of bn1's backward (catseg_dconv3_bnbwd), catseg_bn_backward_pre
    finishes.  dq, dgamma, dbeta against fp64 autograd through relu(batch_norm(q)) -> conv2d, and against the two-pass route
    (plain backward-data + catseg_bn_backward): the masked gradient must be bit-identical (same products, same mask expression)."""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    B, H, W, C = case
    if not lib.catseg_dconv3_supported(C):
        pytest.skip("width not built")
    g = torch.Generator().manual_seed(sum(case) + 1)
    q = torch.randn(B, C, H, W, generator=g) * torch.exp(torch.randn(1, C, 1, 1, generator=g)) + torch.randn(1, C, 1, 1, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, 0.3 * torch.randn(C, generator=g)
    gy = torch.randn(B, C, H, W, generator=g)
    q64, g64, b64 = q.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    out = F.relu(F.batch_norm(q64, None, None, g64, b64, True, 0.1, 1e-5))
    F.conv2d(out, w.double(), None, 1, 1, 1).backward(gy.double())
    qd, gyd = nhwc(q), nhwc(gy)
    if h2:      # the incoming gradient carries its producer's amax record: the two-plane fp16 build of the kernel takes the launch
        gyd._amax = _record_of(ops, gyd)
    gd, bd = gamma.cuda(), beta.cuda()
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    stats, _ = ops.bn_train_stats(qd, gd, 1e-5, 0.1, torch.zeros(C).cuda(), torch.ones(C).cuda())
    saved = (ops.PRECISION, ops.DCONV3_MIN_ROWS)
    ops.PRECISION, ops.DCONV3_MIN_ROWS = "bf16x3", 1
    try:
        ops.PROFILE = []
        # fused: one backward-data launch + merge + apply
        gbuf = torch.full((B, H, W, C), float("nan")).cuda()
        r = ops.conv_bwd_data(gyd, wd, (B, H, W, C), 3, 3, 1, 1, 1, out=gbuf, bn_src=(qd, stats, gd, bd))
        assert isinstance(r, tuple), "the direct kernel did not take the layer"
        dgam, dbet = torch.empty(C).cuda(), torch.empty(C).cuda()
        dq = ops.bn_backward_pre(r[0], qd, stats, gd, r[1], dgam, dbet)
        # two-pass route
        dz = ops.conv_bwd_data(gyd, wd, (B, H, W, C), 3, 3, 1, 1, 1)
        dgam2, dbet2 = torch.empty(C).cuda(), torch.empty(C).cuda()
        dq2 = ops.bn_backward(dz, None, qd, stats, gd, True, dgam2, dbet2, beta=bd)
        kinds = {k for k, *_ in ops.PROFILE}
        assert kinds >= ({"dgrad_d3h"} if h2 and ops.TRUNK == "f16x2" else {"dgrad_d3"}), kinds
    finally:
        ops.PROFILE = None
        ops.PRECISION, ops.DCONV3_MIN_ROWS = saved
        ops.release_b3_cache()
    zpos = nhwc(out.detach().float()) > 0
    # (a pixel whose fp32 z is within rounding of 0 may be masked differently from the fp64 reference; both GPU routes agree exactly)
    zd = torch.addcmul(bd, qd - stats[:C], gd * stats[C:])
    differs = r[0] != torch.where(zd > 0, dz, torch.zeros_like(dz))
    assert not bool((differs & (zd.abs() > 1e-6)).any())       # (torch's own z may round differently from the kernels' fma at |z| ~ 0)
    assert int((zpos != (zd > 0)).sum()) <= 2 + zpos.numel() // 20000
    close(nchw(dq), q64.grad, 5e-5)
    close(dgam, g64.grad, 5e-5)
    close(dbet, b64.grad, 5e-5)
    close(dq, dq2, 1e-5)
    close(dgam, dgam2, 1e-5)
    close(dbet, dbet2, 1e-5)


def _record_of(ops, t, slack=1.0):
    """an amax record as a producing kernel leaves it (max over its slots = bits of max|t|; slack > 1: a conservative bound)"""
    rec = ops.new_amax(t.device)
    rec[32 * 5:32 * 5 + 1] = (t.abs().max() * slack).reshape(1).view(torch.int32)       # (any one of the 16 slots)
    return rec


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("mag", [1.0, 2e-6, 3e4])
def test_dconv3_f16x2_forward_backward_data_vs_fp64(ops, case, spec, mag):
    """the two-plane fp16 build of the direct kernels (csrc/dconv3_f16x2.hip): prescale from the input's amax record, weight image with
    its own exponent; forward (+bias, BatchNorm partials) and backward-data (write / accumulate) against fp64 at activation magnitudes
    fp16 could not hold unscaled; a conservative record (8x the true maximum) changes nothing measurable"""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    B, H, W, C = case
    if not lib.catseg_dconv3_supported(C) or (mag != 1.0 and (B * H * W < 100 or C == 64)):
        pytest.skip("width not built / covered at mag 1")
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, C, 1, 1, generator=g)) * mag
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5
    b = torch.randn(C, generator=g) * mag
    xr = x.double().requires_grad_()
    y64 = F.conv2d(xr, w.double(), b.double(), 1, 1, 1)
    gy = torch.randn(y64.shape, generator=g) * 1e-5
    y64.backward(gy.double())
    xd, wd = nhwc(x), w.cuda().contiguous(memory_format=torch.channels_last)
    wimg = ops.dconv3_weight_image(wd, h2=True)
    y, part = ops.dconv3(xd, wimg, b.cuda(), bn_stats=True, x_amax=_record_of(ops, xd))
    e = close2(nchw(y), y64.detach(), 2e-5)
    y8, _ = ops.dconv3(xd, wimg, b.cuda(), bn_stats=True, x_amax=_record_of(ops, xd, 8.0))
    close2(nchw(y8), y64.detach(), 2e-5)
    gamma, rm, rv = torch.ones(C).cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda()
    stats, _ = ops.bn_finalize(part, B * H * W, C, gamma, 0.0, 0.1, rm, rv)
    y2 = y64.detach().permute(1, 0, 2, 3).reshape(C, -1)
    assert float((stats[:C].cpu().double() - y2.mean(1)).abs().max()) <= 2e-5 * float(y2.abs().max())
    if B * H * W > 1:
        assert float((stats[C:].cpu().double() * y2.var(1, unbiased=False).sqrt() - 1).abs().max()) <= 1e-4
    wimg_t = ops.dconv3_weight_image(wd, backward_data=True, h2=True)
    gyd = nhwc(gy)
    dx = ops.dconv3(gyd, wimg_t, x_amax=_record_of(ops, gyd))
    e2 = close2(nchw(dx), xr.grad, 2e-5)
    dx2 = ops.dconv3(gyd, wimg_t, out=dx.clone(), accumulate=True, x_amax=_record_of(ops, gyd))
    close2(nchw(dx2), 2 * xr.grad, 2e-5)
    ops.release_b3_cache()
    print("dconv3 f16x2 %s x %g: forward %.2g, backward-data %.2g of the output scale" % (case, mag, e, e2))


def close2(a, b, rtol):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-300
    assert err <= rtol * scale, "max abs err %g (ref scale %g)" % (err, scale)
    return err / scale


def test_dconv3_inside_concat_buffers(ops):
    """input and output as channel slices of wider buffers (row strides > C); neighbours of the output slice stay untouched"""
    B, H, W, C = 2, 11, 21, 48
    g = torch.Generator().manual_seed(5)
    xin = torch.randn(B, H, W, 112, generator=g).cuda()
    x = xin[..., 16:16 + C]
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    ybuf = torch.full((B, H, W, 96), 7.0).cuda()
    y = ybuf[..., 32:32 + C]
    ops.dconv3(x, ops.dconv3_weight_image(w), None, out=y)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1, 1).permute(0, 2, 3, 1)
    close(y, ref, 2e-5)
    assert float((ybuf[..., :32] - 7.0).abs().max()) == 0.0
    assert float((ybuf[..., 32 + C:] - 7.0).abs().max()) == 0.0
    ops.release_b3_cache()


def test_conv_dispatch_uses_direct_kernel(ops):
    """ops.conv_fwd / conv_bwd_data route eligible layers to the direct kernel and agree with the fp32 implicit GEMM"""
    B, H, W, C = 2, 24, 40, 48
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
    saved = (ops.DCONV3_MIN_ROWS, ops.PRECISION)
    try:
        ops.PRECISION, ops.DCONV3_MIN_ROWS = "bf16x3", 1
        ops.PROFILE = []
        y, part = ops.conv_fwd(x, w, None, C, 3, 3, 1, 1, 1, bn_stats=True)
        dx = ops.conv_bwd_data(y, w, tuple(x.shape), 3, 3, 1, 1, 1)
        kinds = [k for k, *_ in ops.PROFILE]
        assert "fwd_d3" in kinds and "dgrad_d3" in kinds and len(part) == 4
        ops.PROFILE = None
        ops.PRECISION = "fp32"
        y0 = ops.conv_fwd(x, w, None, C, 3, 3, 1, 1, 1)
        dx0 = ops.conv_bwd_data(y, w, tuple(x.shape), 3, 3, 1, 1, 1)
        close(y, y0, 1e-5)
        close(dx, dx0, 1e-5)
    finally:
        ops.PROFILE = None
        ops.DCONV3_MIN_ROWS, ops.PRECISION = saved
        ops.release_b3_cache()


def test_batched_weight_images_equal_single_prep(ops):
    """ops.Dconv3Bank (one launch over the flat parameter buffer) writes the images catseg_dconv3_prep writes layer by layer"""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    g = torch.Generator().manual_seed(3)
    widths = [48, 96, 48, 192]
    sizes = [c * c * 9 for c in widths]
    offs, o = [], 7 * 64
    for n in sizes:
        offs.append(o)
        o += (n + 63) // 64 * 64 + 64
    flat = torch.randn(o, generator=g).cuda()
    ws = [flat[a:a + n].view(c, 3, 3, c).permute(0, 3, 1, 2) for a, n, c in zip(offs, sizes, widths)]
    bank = ops.Dconv3Bank(flat, list(zip(ws, offs)))
    ops.release_b3_cache()
    bank.refresh()
    got = {k: v.clone() for k, v in ops._d3_wimg.items()}
    ops.release_b3_cache()
    for w in ws:
        for dg in (False, True):
            assert torch.equal(got[(w.data_ptr(), dg, False)], ops.dconv3_weight_image(w, backward_data=dg))
    ops.release_b3_cache()
    # the two-plane fp16 bank: every image = fp16 split of w * 2^e with the layer's own exponent e = 14 - floor(log2 max|w|)
    bank = ops.Dconv3Bank(flat, list(zip(ws, offs)), h2=True)
    bank.refresh()
    for w, c in zip(ws, widths):
        img, rec = ops._d3_wimg[(w.data_ptr(), False, True)]
        amax = float(w.abs().max())
        e = int(rec.cpu()[1])
        assert 2.0 ** 14 <= amax * 2.0 ** e < 2.0 ** 15 and img.numel() == lib.catseg_dconv3_f16x2_wimg_bytes(c)
        v = img.view(torch.float16).float()
        assert float(v.abs().max()) <= 2.0 ** 15 and bool(torch.isfinite(v).all())
    ops.release_b3_cache()


WG_CASES = [(2, 8, 32, 48), (1, 19, 37, 48), (3, 5, 7, 48), (1, 1, 50, 48), (2, 33, 1, 48),
            (2, 8, 32, 96), (1, 19, 37, 96), (2, 3, 5, 96), (1, 11, 21, 192), (1, 17, 30, 384), (2, 8, 32, 64), (1, 19, 37, 64)]


@pytest.mark.parametrize("case", WG_CASES)
@pytest.mark.parametrize("mags", [(1.0, 1.0), (3e3, 2e-6)])
def test_dwgrad3_f16x2_vs_fp64(ops, case, mags):
    """the two-plane fp16 build of the direct backward-weight kernel: both operands prescaled from their amax records"""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    B, H, W, C = case
    if not lib.catseg_dwgrad3_supported(C):
        pytest.skip("width not built")
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, C, 1, 1, generator=g)) * mags[0]
    dy = torch.randn(B, C, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, C, 1, 1, generator=g)) * mags[1]
    w = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1, 1).backward(dy.double())
    xd, dyd = nhwc(x), nhwc(dy)
    xd._amax, dyd._amax = _record_of(ops, xd), _record_of(ops, dyd, 3.0)
    dw = torch.full((C, C, 3, 3), float("nan")).cuda().contiguous(memory_format=torch.channels_last)
    saved = ops.TRUNK, ops.PRECISION
    ops.TRUNK, ops.PRECISION = "f16x2", "bf16x3"
    try:
        ops.PROFILE = []
        ops.dwgrad3(xd, dyd, dw)
        assert [k for k, *_ in ops.PROFILE] == ["wgrad_d3h"]
    finally:
        ops.PROFILE = None
        ops.TRUNK, ops.PRECISION = saved
    close2(dw, w.grad, 2e-5)


@pytest.mark.parametrize("case", WG_CASES)
def test_dwgrad3_vs_fp64(ops, case):
    """direct split-precision backward-weight (csrc/dwgrad3_b3.hip) against the fp64 autograd of F.conv2d, inputs with a wide dynamic
    range, tensors inside wider buffers, every block count from one tile per block to one block"""
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    B, H, W, C = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    x = torch.randn(B, C, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, C, 1, 1, generator=g))
    gy = torch.randn(B, C, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, C, 1, 1, generator=g))
    wr = torch.zeros(C, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wr, None, 1, 1, 1).backward(gy.double())
    xbuf = torch.zeros(B, H, W, C + 16).cuda()
    xbuf[..., 8:8 + C] = nhwc(x)
    xd = xbuf[..., 8:8 + C]
    gyd = nhwc(gy)
    try:
        for blocks in (0, 1, 24, 100000):
            lib.catseg_debug_set_dwgrad3_blocks(blocks)
            dw = torch.full((C, C, 3, 3), 3.0).cuda().contiguous(memory_format=torch.channels_last)
            db = torch.empty(C).cuda()
            ops.dwgrad3(xd, gyd, dw, db)
            close(dw, wr.grad, 2e-5)
            close(db, gy.double().sum((0, 2, 3)), 2e-5)
    finally:
        lib.catseg_debug_set_dwgrad3_blocks(0)
