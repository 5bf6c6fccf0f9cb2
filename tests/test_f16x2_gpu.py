"""Two-plane fp16 split precision (csrc/igemm_f16x2.hip; call sites: the OCR / auxiliary head convolutions of models/OCR.py:88-104 and
UPerNet's fusion layers): the split itself (power-of-two prescale from the tensor's amax, h + l = x 2^e to 22 bits), forward
(+ bias, zero pad columns, BatchNorm partials, fused inference epilogue) and backward-data against fp64 F.conv2d -- on inputs whose
magnitude fp16 could not hold unscaled (gradients of 1e-7, activations of 1e+6), with a wide dynamic range across channels."""
import numpy as np
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


def ohwi(w):
    return w.cuda().contiguous(memory_format=torch.channels_last)


def close(a, b, rtol):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-300
    assert err <= rtol * scale, "max abs err %g (ref scale %g): %g relative" % (err, scale, err / scale)
    return err / scale


def planes_to_float(planes, scale, rows, C):
    """(blocked fp16 planes, {amax bits, e}) -> fp64 [rows, C] of (h + l) * 2^-e"""
    h = planes[0].cpu().view(torch.float16).double()       # [C16, rows, 16]
    l = planes[1].cpu().view(torch.float16).double()
    v = (h + l).permute(1, 0, 2).reshape(rows, -1)[:, :C]
    e = int(scale.cpu()[1])
    return v * 2.0 ** (-e), e


@pytest.mark.parametrize("mag", [1.0, 3e-7, 4e6, 1e-25, 1e25])
def test_split_reconstructs_22_bits_at_any_magnitude(ops, mag):
    g = torch.Generator().manual_seed(3)
    rows, C = 300, 40                                     # ragged: rows % 64 != 0, C % 16 != 0
    x = (torch.randn(rows, C, generator=g) * torch.exp(2 * torch.randn(1, C, generator=g)) * mag).float()
    xb = torch.zeros(rows, 48)
    xb[:, :C] = x
    xd = xb.cuda().view(1, rows, 1, 48)[..., :C]            # row stride 48 > C
    planes, scale = ops.split2h_blocked(xd)
    v, e = planes_to_float(planes, scale, rows, C)
    amax = float(x.abs().max())
    assert np.frombuffer(np.int32(int(scale.cpu()[0])).tobytes(), dtype=np.float32)[0] == np.float32(amax)
    assert 2.0 ** 14 <= amax * 2.0 ** e < 2.0 ** 15
    xr = x.double()
    err = (v - xr).abs()
    # 22 bits where l is a normal fp16 (|x 2^e| >= 2^-3), never worse than 2^-25 of the scaled unit = 2^-39 of the largest element
    bound = torch.maximum(xr.abs() * 2.0 ** -22, torch.full_like(xr, 2.0 ** -25 * 2.0 ** -e))
    assert bool((err <= bound).all()), float((err / bound).max())
    # the channel tail of the last 16-channel chunk is zero
    tail = (planes.cpu().view(torch.float16)[:, -1, :, C - 32:].abs().sum())
    assert float(tail) == 0.0


def test_split_of_zeros_and_nan(ops):
    z = torch.zeros(1, 4, 4, 16).cuda()
    planes, scale = ops.split2h_blocked(z)
    assert int(scale.cpu()[1]) == 0 and float(planes.cpu().view(torch.float16).abs().sum()) == 0.0
    z[0, 1, 2, 3] = float("nan")
    planes, scale = ops.split2h_blocked(z)
    assert int(scale.cpu()[1]) == 0 and bool(torch.isnan(planes[0].cpu().view(torch.float16)).any())


# B, H, W, Cin, Cout, k, stride, pad, dil, x magnitude, w magnitude
CASES = [(2, 20, 24, 64, 256, 3, 1, 1, 1, 1.0, 1.0),
         (1, 33, 17, 96, 200, 3, 1, 1, 1, 3e-7, 1.0),        # gradient-sized activations, ragged N
         (2, 9, 40, 128, 512, 1, 1, 0, 1, 4e6, 2e-5),        # 1x1, huge activations, tiny weights
         (1, 24, 24, 64, 320, 3, 2, 1, 1, 1.0, 1.0),         # stride 2 (forward only)
         (1, 30, 30, 48, 256, 3, 1, 2, 2, 1.0, 50.0)]        # dilation 2


@pytest.mark.parametrize("case", CASES)
def test_forward_and_backward_data_vs_fp64(ops, case):
    B, H, W, Ci, Co, k, s, p, d, xm, wm = case
    g = torch.Generator().manual_seed(sum(int(v) for v in case[:9]))
    x = torch.randn(B, Ci, H, W, generator=g) * torch.exp(1.5 * torch.randn(1, Ci, 1, 1, generator=g)) * xm
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5 * wm
    b = torch.randn(Co, generator=g) * xm * wm
    xr = x.double().requires_grad_()
    y64 = F.conv2d(xr, w.double(), b.double(), s, p, d)
    gy = torch.randn(y64.shape, generator=g) * 1e-6
    y64.backward(gy.double())
    xd, wd = nhwc(x), ohwi(w)
    xp, xs = ops.split2h_blocked(xd)
    wp, ws = ops.split2h_weight_blocked(wd)
    import ctypes
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    Ho, Wo = y64.shape[2:]
    ld = (Co + 8 + 31) // 32 * 32
    out = torch.full((B, Ho, Wo, ld), 7.0).cuda()
    yv = out[..., :Co]
    dsc = ops.make_desc(xd.shape, Ci, Co, ld, k, k, s, p, d)
    nt_max = (B * Ho * Wo + 255) // 256
    part = torch.empty(3 * nt_max * Co).cuda()
    tr, nt = ctypes.c_int(0), ctypes.c_int(0)
    ops.check(lib.catseg_conv2d_fwd_f16x2_blocked(ctypes.byref(dsc), ops.ptr(xp), ops.ptr(xs), ops.ptr(wp), ops.ptr(ws), ops.ptr(b.cuda()),
                                                  ops.ptr(yv), Co + 8, ops.ptr(part), part.numel(), ctypes.byref(tr), ctypes.byref(nt),
                                                  ops.stream()))
    e = close(nchw(yv), y64.detach(), 2e-5)
    assert float(out[..., Co:Co + 8].abs().max()) == 0.0
    if ld > Co + 8:
        assert float((out[..., Co + 8:] - 7.0).abs().max()) == 0.0
    # BatchNorm statistics from the epilogue's partials
    assert tr.value == 256 and nt.value == nt_max
    gamma, rm, rv = torch.ones(Co).cuda(), torch.zeros(Co).cuda(), torch.ones(Co).cuda()
    stats, _ = ops.bn_finalize((part, nt.value, tr.value), B * Ho * Wo, Co, gamma, 0.0, 0.1, rm, rv)
    y2 = y64.detach().permute(1, 0, 2, 3).reshape(Co, -1)
    assert float((stats[:Co].cpu().double() - y2.mean(1)).abs().max()) <= 2e-5 * float(y2.abs().max())
    assert float((stats[Co:].cpu().double() * y2.var(1, unbiased=False).sqrt() - 1).abs().max()) <= 1e-4
    if s == 1:
        gyd = nhwc(gy)
        gp, gs = ops.split2h_blocked(gyd)
        wtp, wts = ops.split2h_weight_t_blocked(wd)
        dx = torch.full((B, H, W, Ci), float("nan")).cuda()
        dsc = ops.make_desc(xd.shape, Ci, Co, (Co + 7) // 8 * 8, k, k, s, p, d)
        ops.check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(dsc), ops.ptr(gp), ops.ptr(gs), ops.ptr(wtp), ops.ptr(wts), ops.ptr(dx), 0,
                                                           ops.stream()))
        e2 = close(nchw(dx), xr.grad, 2e-5)
        ops.check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(dsc), ops.ptr(gp), ops.ptr(gs), ops.ptr(wtp), ops.ptr(wts), ops.ptr(dx), 1,
                                                           ops.stream()))
        close(nchw(dx), 2 * xr.grad, 2e-5)
        print("f16x2 %s: forward %.2g, backward-data %.2g of the output scale" % (case, e, e2))
    # backward-weight from the planar planes (written by the same pass as the blocked ones)
    w64 = w.double().requires_grad_()
    F.conv2d(x.double(), w64, None, s, p, d).backward(gy.double())
    gyd = nhwc(gy)
    xb, xpl, xsc = ops.split2h(xd, blocked=True, planar=True)
    assert torch.equal(xb, xp)
    _, gpl, gsc = ops.split2h(gyd, blocked=False, planar=True)
    dsc = ops.make_desc(xd.shape, Ci, Co, (Co + 7) // 8 * 8, k, k, s, p, d)
    ws = torch.empty(max(lib.catseg_conv2d_bwd_weight_f16x2_workspace(ctypes.byref(dsc)), 256), dtype=torch.uint8).cuda()
    dw = torch.full((Co, Ci, k, k), float("nan")).cuda().contiguous(memory_format=torch.channels_last)
    ops.check(lib.catseg_conv2d_bwd_weight_f16x2(ctypes.byref(dsc), ops.ptr(xpl), ops.ptr(xsc), ops.ptr(gpl), ops.ptr(gsc), ops.ptr(dw),
                                                 ops.ptr(ws), ws.numel(), ops.stream()))
    e3 = close(dw, w64.grad, 2e-5)
    print("f16x2 %s: backward-weight %.2g of the output scale" % (case, e3))
    # ... and from the BLOCKED planes (the operand layout of the forward / backward-data kernels): the same values reach the same MFMAs in the
    # same order -- bit-identical results
    gb = ops.split2h_blocked(gyd)[0]
    dwb = torch.full((Co, Ci, k, k), float("nan")).cuda().contiguous(memory_format=torch.channels_last)
    ops.check(lib.catseg_conv2d_bwd_weight_f16x2_blocked(ctypes.byref(dsc), ops.ptr(xb), ops.ptr(xsc), ops.ptr(gb), ops.ptr(gsc), ops.ptr(dwb),
                                                         ops.ptr(ws), ws.numel(), ops.stream()))
    assert torch.equal(dwb, dw), "backward-weight from blocked planes differs from the planar-plane result"


@pytest.mark.parametrize("case", [(2, 32, 32, 400, 512, 1), (1, 32, 48, 176, 256, 3), (1, 16, 16, 720, 512, 1)])
def test_fast_epilogue_bit_identical_to_the_general_one(ops, case):
    """igemm_h2w8_kernel<FAST> (launches whose row tiles are all full: M % 256 == 0, no residual, no pad columns) against the general,
    element-predicated epilogue of the same kernel (catseg_debug_set_h2w_slow_epilogue): forward with bias + BatchNorm partials, backward-data
    into a RAGGED column count (N = Cin not a multiple of 256: the HRNet head's 720) with and without accumulation -- bit for bit; and the
    forward against fp64."""
    import ctypes
    from miccai2021_cataract_semantic_segmentation_amd._lib import lib
    B, H, W, Ci, Co, k = case
    assert (B * H * W) % 256 == 0
    g = torch.Generator().manual_seed(B + H + W + Ci + Co + k)
    x = torch.randn(B, Ci, H, W, generator=g) * torch.exp(1.0 * torch.randn(1, Ci, 1, 1, generator=g))
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    b = torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, H, W, generator=g) * 1e-6
    xd, wd, gyd = nhwc(x), ohwi(w), nhwc(gy)
    xp, xs = ops.split2h_blocked(xd)
    wp, ws = ops.split2h_weight_blocked(wd)
    gp, gs = ops.split2h_blocked(gyd)
    wtp, wts = ops.split2h_weight_t_blocked(wd)
    base = torch.randn(B, H, W, Ci, generator=g).cuda() * 1e-6
    res = []
    try:
        for slow in (0, 1):
            lib.catseg_debug_set_h2w_slow_epilogue(slow)
            y = torch.full((B, H, W, Co), float("nan")).cuda()
            dsc = ops.make_desc(xd.shape, Ci, Co, Co, k, k, 1, k // 2, 1)
            part = torch.full((3 * (B * H * W // 256) * Co,), float("nan")).cuda()
            tr, nt = ctypes.c_int(0), ctypes.c_int(0)
            ops.check(lib.catseg_conv2d_fwd_f16x2_blocked(ctypes.byref(dsc), ops.ptr(xp), ops.ptr(xs), ops.ptr(wp), ops.ptr(ws), ops.ptr(b.cuda()),
                                                          ops.ptr(y), 0, ops.ptr(part), part.numel(), ctypes.byref(tr), ctypes.byref(nt), ops.stream()))
            dx0 = torch.full((B, H, W, Ci), float("nan")).cuda()
            dsc = ops.make_desc(xd.shape, Ci, Co, (Co + 7) // 8 * 8, k, k, 1, k // 2, 1)
            ops.check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(dsc), ops.ptr(gp), ops.ptr(gs), ops.ptr(wtp), ops.ptr(wts), ops.ptr(dx0), 0,
                                                               ops.stream()))
            dx1 = base.clone()
            ops.check(lib.catseg_conv2d_bwd_data_f16x2_blocked(ctypes.byref(dsc), ops.ptr(gp), ops.ptr(gs), ops.ptr(wtp), ops.ptr(wts), ops.ptr(dx1), 1,
                                                               ops.stream()))
            torch.cuda.synchronize()
            res.append((y, part, dx0, dx1))
    finally:
        lib.catseg_debug_set_h2w_slow_epilogue(0)
    for a, bb in zip(res[0], res[1]):
        assert torch.equal(a, bb)
    y64 = F.conv2d(x.double(), w.double(), b.double(), 1, k // 2)
    close(nchw(res[0][0]), y64, 2e-5)


def test_dispatch_takes_the_f16x2_kernels_and_matches_bf16x3(ops):
    """ops.conv_fwd / conv_bwd_data on a head-shaped layer: the f16x2 kernels run by default (CATSEG_HEADS), and agree with the
    six-product bf16x3 path to fp32 rounding"""
    g = torch.Generator().manual_seed(11)
    B, H, W, Ci, Co = 2, 24, 40, 208, 256
    x = nhwc(torch.randn(B, Ci, H, W, generator=g))
    w = ohwi(torch.randn(Co, Ci, 3, 3, generator=g) * 0.04)
    dy = nhwc(torch.randn(B, Co, H, W, generator=g) * 1e-5)
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.HEADS)
    res = {}
    try:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = "bf16x3", 1, 16, 16, 1, 1
        for heads in ("f16x2", "bf16x3"):
            ops.HEADS = heads
            ops.release_b3_cache()
            ops.PROFILE = []
            y, part = ops.conv_fwd(x, w, None, Co, 3, 3, 1, 1, 1, bn_stats=True, train=True)
            dw = torch.empty_like(w)
            ops.conv_bwd_weight(x, dy, dw, None, 3, 3, 1, 1, 1)
            dx = ops.conv_bwd_data(dy, w, tuple(x.shape), 3, 3, 1, 1, 1)
            kinds = [q[0] for q in ops.PROFILE]
            ops.PROFILE = None
            want = ("fwd_h2", "dgrad_h2", "wgrad_h2") if heads == "f16x2" else ("fwd_b3", "dgrad_b3", "wgrad_b3")
            assert all(k in kinds for k in want), kinds
            assert part is not None
            res[heads] = (y.clone(), dx.clone(), dw.clone())
    finally:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.HEADS = saved
        ops.PROFILE = None
        ops.release_b3_cache()
    close(res["f16x2"][0], res["bf16x3"][0], 5e-6)      # (each is ~1e-6 from fp64: fp32 accumulation over K = 1872)
    close(res["f16x2"][1], res["bf16x3"][1], 5e-6)
    close(res["f16x2"][2], res["bf16x3"][2], 5e-6)
