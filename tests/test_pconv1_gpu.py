"""csrc/pconv1.hip: pointwise (1 x 1) convolutions in split precision with the fp32 -> 2 x fp16 split in registers -- forward, backward-data
(the same kernel on the transposed weight image) and backward-weight against float64 matrix products (the arithmetic of F.conv2d with a
1 x 1 filter, which is what the oracle's networks call: oracle/nets.py: conv), on the layer shapes of the HRNet-W48 / OCR step
(models/HRNetv2.py:68-106,237-261, models/OCR.py:186-235 of the reference) with ragged row counts, padded row strides, bias, accumulation
and the BatchNorm partials of the epilogue."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5      # of the output scale, against float64 -- the bar of every split-precision kernel (tests/test_f16x2_gpu.py)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _rec(t, slack=1.0):
    """an amax record as a producing kernel leaves it: the bits of max|t| (times slack: a bound) in one of the 16 slots"""
    from miccai2021_cataract_semantic_segmentation_amd import ops
    rec = ops.new_amax(t.device)
    m = (t.detach().abs().max() * slack).float().reshape(1)
    rec[32 * 5] = m.view(torch.int32)[0]
    return rec


def _act(rows, C, ld, gen, scale=1.0, spread=3.0):
    """activation-like data: per-channel magnitudes over e^+-spread, a few exact zeros (ReLU)"""
    x = torch.randn(rows, C, generator=gen, dtype=torch.float64) * torch.exp(spread * (2 * torch.rand(C, generator=gen, dtype=torch.float64) - 1)) * scale
    x[torch.rand(rows, C, generator=gen) < 0.2] = 0.0
    buf = torch.full((rows, ld), float("nan"), dtype=torch.float32)
    buf[:, :C] = x.float()
    return buf.cuda()[:, :C], buf[:, :C].double()


SHAPES = [  # (rows, K = Cin, N = Cout)
    (4096 + 37, 64, 256), (2049, 256, 64), (3000, 64, 64), (2500, 512, 256), (2304, 256, 256), (2200, 256, 512),
    (2100, 96, 48), (2050, 192, 96), (2060, 384, 48), (2070, 384, 192), (2111, 192, 48), (5000, 48, 96), (2300, 1024, 512),
    (2150, 256, 1024), (2090, 1024, 256), (2077, 512, 2048), (2310, 2048, 512), (2081, 128, 520)]     # (wide layers: ResNet bottlenecks, > 2 column tiles)


@pytest.mark.parametrize("rows,K,N", SHAPES)
def test_forward_and_backward_data_vs_float64(rows, K, N):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    g = torch.Generator().manual_seed(rows + K + N)
    x, x64 = _act(rows, K, K + 8, g, scale=3e-3)
    w = (torch.randn(N, K, 1, 1, generator=g) * 0.07).cuda().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(N, generator=g).cuda()
    x._amax = _rec(x, 1.7)
    x4 = x.view(1, 1, rows, K) if x.is_contiguous() else torch.as_strided(x, (1, 1, rows, K), (rows * (K + 8), rows * (K + 8), K + 8, 1))
    x4._amax = x._amax
    # forward + bias + BatchNorm partials
    out = torch.full((1, 1, rows, N + 4), float("nan"), dtype=torch.float32, device="cuda")[..., :N]
    y, part = ops.pconv1(x4, ops.p1_weight_image(w), bias, N, out, bn_stats=True)
    ref = x64 @ w.double().cpu().view(N, K).t() + bias.double().cpu()
    got = y.cpu().double().view(rows, N)
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= TOL * scale, (float((got - ref).abs().max()), scale)
    assert part is not None and part[1] == (rows + 255) // 256 and part[2] == 256
    stats, _ = ops.bn_finalize(part, rows, N, torch.ones(N, device="cuda"), 1e-5, 0.1, torch.zeros(N, device="cuda"), torch.ones(N, device="cuda"))
    mean, invstd = stats[:N].cpu().double(), stats[N:].cpu().double()
    assert torch.allclose(mean, ref.mean(0), atol=2e-6 * scale)
    var = ref.var(0, unbiased=False)
    assert torch.allclose(1.0 / (invstd * invstd) - 1e-5, var, rtol=2e-4, atol=1e-9 * scale * scale)
    # backward-data = the same kernel on dy with the transposed image, accumulating into an existing gradient
    if not ops.lib.catseg_pconv1_supported(K, N):
        ops.release_b3_cache()
        return
    dy, dy64 = _act(rows, N, N, g, scale=2e-6, spread=2.0)
    dy4 = dy.view(1, 1, rows, N)
    dy4._amax = _rec(dy)
    base = torch.randn(1, 1, rows, K, generator=g).cuda() * 1e-7
    dx = base.clone()
    ops.pconv1(dy4, ops.p1_weight_image(w, transposed=True), None, K, dx, accumulate=True)
    refd = dy64 @ w.double().cpu().view(N, K)
    gotd = (dx.cpu().double() - base.cpu().double()).view(rows, K)
    sd = float(refd.abs().max())
    assert float((gotd - refd).abs().max()) <= TOL * sd + 2e-7 * float(base.abs().max()), (float((gotd - refd).abs().max()), sd)
    ops.release_b3_cache()


@pytest.mark.parametrize("rows,K,N", SHAPES)
def test_backward_weight_vs_float64(rows, K, N):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    if not ops.lib.catseg_pconv1_wgrad_supported(N, K):
        pytest.skip("shape outside the backward-weight kernel")
    g = torch.Generator().manual_seed(7 * rows + K + N)
    rows = rows * 3 + 5
    x, x64 = _act(rows, K, K + 4, g, scale=0.5)
    dy, dy64 = _act(rows, N, N, g, scale=3e-6, spread=2.0)
    x._amax, dy._amax = _rec(x, 1.3), _rec(dy)
    dw = torch.full((N, K), float("nan"), dtype=torch.float32, device="cuda")
    need = ops.lib.catseg_pconv1_wgrad_workspace(rows, N, K)
    ws = torch.empty(need + 256, dtype=torch.uint8, device="cuda")
    ops.check(ops.lib.catseg_pconv1_wgrad(rows, N, K, ops.ptr(dy), ops.ld_of(dy), ops.ptr(dy._amax), ops.ptr(x), ops.ld_of(x), ops.ptr(x._amax),
                                          ops.ptr(dw), ops.ptr(ws), need, ops.stream()))
    ref = dy64.t() @ x64
    got = dw.cpu().double()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= TOL * scale, (float((got - ref).abs().max()), scale)
    # deterministic: a second launch gives the same bits
    dw2 = torch.empty_like(dw)
    ops.check(ops.lib.catseg_pconv1_wgrad(rows, N, K, ops.ptr(dy), ops.ld_of(dy), ops.ptr(dy._amax), ops.ptr(x), ops.ld_of(x), ops.ptr(x._amax),
                                          ops.ptr(dw2), ops.ptr(ws), need, ops.stream()))
    assert torch.equal(dw, dw2)


def test_conv_wrappers_take_the_pointwise_route_and_match_the_fp32_kernels():
    """ops.conv_fwd / conv_bwd_data / conv_bwd_weight pick csrc/pconv1.hip for a 1 x 1 layer whose operands carry amax records (and the
    fp32 MFMA kernels without them); both agree to the split-precision tolerance"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    if not ops._trunk_h2():
        pytest.skip("the pointwise route needs the f16x2 trunk arithmetic")
    g = torch.Generator().manual_seed(3)
    B, H, W, Cin, Cout = 2, 40, 56, 64, 256
    x = torch.randn(B, H, W, Cin, generator=g).relu().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1).cuda().contiguous(memory_format=torch.channels_last)
    dy = (torch.randn(B, H, W, Cout, generator=g) * 1e-5).cuda()
    prof, ops.PROFILE = ops.PROFILE, []
    saved = (ops.P1_MIN_ROWS, ops.P1_WGRAD_MIN_DIM)
    ops.P1_MIN_ROWS, ops.P1_WGRAD_MIN_DIM = 1, 1
    try:
        y0 = ops.conv_fwd(x, w, None, Cout, 1, 1)
        dx0 = ops.conv_bwd_data(dy, w, tuple(x.shape), 1, 1)
        dw0 = ops.conv_bwd_weight(x, dy, torch.empty_like(w), None, 1, 1)
        kinds0 = [k[0] for k in ops.PROFILE]
        ops.PROFILE = []
        x._amax, dy._amax = _rec(x), _rec(dy)
        y1 = ops.conv_fwd(x, w, None, Cout, 1, 1)
        dx1 = ops.conv_bwd_data(dy, w, tuple(x.shape), 1, 1)
        dw1 = ops.conv_bwd_weight(x, dy, torch.empty_like(w), None, 1, 1)
        kinds1 = [k[0] for k in ops.PROFILE]
    finally:
        ops.PROFILE = prof
        ops.P1_MIN_ROWS, ops.P1_WGRAD_MIN_DIM = saved
        ops.release_b3_cache()
    torch.cuda.synchronize()
    assert kinds0 == ["fwd", "dgrad", "wgrad"] and kinds1 == ["fwd_p1", "dgrad_p1", "wgrad_p1"]
    for a, b in ((y0, y1), (dx0, dx1), (dw0, dw1)):
        s = float(a.abs().max())
        assert float((a - b).abs().max()) <= 4e-5 * s


CONVS = [  # (B, H, W, Cin, Cout, k, stride, pad, dil)
    (2, 34, 50, 48, 96, 3, 2, 1, 1), (2, 33, 47, 48, 48, 3, 2, 1, 1), (1, 40, 64, 256, 48, 3, 1, 1, 1), (2, 36, 36, 64, 64, 3, 2, 1, 1),
    (2, 20, 30, 96, 192, 3, 2, 1, 1), (1, 24, 24, 192, 384, 3, 2, 1, 1), (1, 30, 30, 64, 128, 3, 1, 2, 2), (1, 26, 38, 96, 96, 3, 2, 1, 1),
    (1, 21, 35, 256, 96, 3, 2, 1, 1), (2, 16, 16, 48, 384, 3, 2, 1, 1),
    # row counts that are multiples of 256 in forward AND in every backward-data parity class: the full-tile epilogue with ragged column
    # counts (48 of a 64-wide tile), gathered output rows and accumulation
    (2, 32, 64, 48, 96, 3, 2, 1, 1), (1, 32, 32, 48, 48, 3, 2, 1, 1), (1, 32, 32, 64, 128, 3, 1, 2, 2), (4, 32, 32, 96, 192, 3, 2, 1, 1)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,dil", CONVS)
def test_gather_launches_vs_float64_conv2d(B, H, W, Cin, Cout, k, stride, pad, dil):
    """the gather launches of csrc/pconv1.hip (3x3 / stride 2 fuse and transition layers of models/HRNetv2.py:176-198,237-261, the
    256 -> 48 transition, a dilated layer) through ops.conv_fwd / conv_bwd_data / conv_bwd_weight against float64 F.conv2d and its autograd:
    forward with BatchNorm partials, backward-data per input-pixel parity class with accumulation, backward-weight"""
    _need_gpu()
    import torch.nn.functional as F
    from miccai2021_cataract_semantic_segmentation_amd import ops
    if not ops._trunk_h2():
        pytest.skip("needs the f16x2 trunk arithmetic (amax records)")
    g = torch.Generator().manual_seed(B * H + W + Cin + Cout)
    x = (torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64) * torch.exp(2 * torch.rand(1, Cin, 1, 1, generator=g, dtype=torch.float64) - 1)).relu() * 3e-2
    w = torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) * 0.05
    xr, wr = x.float().double().requires_grad_(), w.float().double().requires_grad_()
    yr = F.conv2d(xr, wr, None, stride, pad, dil)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64) * 2e-6
    gy = gy.float().double()
    yr.backward(gy)
    xd = x.float().permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.float().cuda().contiguous(memory_format=torch.channels_last)
    dyd = gy.float().permute(0, 2, 3, 1).contiguous().cuda()
    xd._amax, dyd._amax = _rec(xd, 1.5), _rec(dyd)
    saved = (ops.G1_MIN_ROWS, ops.PROFILE, ops.G1_DGRAD_MIN_CIN)
    ops.G1_MIN_ROWS, ops.PROFILE, ops.G1_DGRAD_MIN_CIN = 1, [], 1
    try:
        y, part = ops.conv_fwd(xd, wd, None, Cout, k, k, stride, pad, dil, bn_stats=True)
        base = torch.randn(xd.shape, generator=g).cuda() * 1e-8
        dx = base.clone()
        ops.conv_bwd_data(dyd, wd, tuple(xd.shape), k, k, stride, pad, dil, out=dx, accumulate=True)
        dw = ops.conv_bwd_weight(xd, dyd, torch.full_like(wd, float("nan")), None, k, k, stride, pad, dil)
        kinds = [q[0] for q in ops.PROFILE]
    finally:
        ops.G1_MIN_ROWS, ops.PROFILE, ops.G1_DGRAD_MIN_CIN = saved
        ops.release_b3_cache()
    torch.cuda.synchronize()
    assert kinds == ["fwd_s2p", "dgrad_s2p", "wgrad_s2p"], kinds
    ref = yr.detach().permute(0, 2, 3, 1)
    s = float(ref.abs().max())
    assert float((y.cpu().double() - ref).abs().max()) <= TOL * s, (float((y.cpu().double() - ref).abs().max()), s)
    rows = ref.shape[0] * ref.shape[1] * ref.shape[2]
    stats, _ = ops.bn_finalize(part, rows, Cout, torch.ones(Cout, device="cuda"), 1e-5, 0.1, torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda"))
    assert torch.allclose(stats[:Cout].cpu().double(), ref.reshape(-1, Cout).mean(0), atol=2e-6 * s)
    refx = xr.grad.permute(0, 2, 3, 1)
    sx = float(refx.abs().max())
    gotx = dx.cpu().double() - base.cpu().double()
    assert float((gotx - refx).abs().max()) <= TOL * sx + 2e-7 * float(base.abs().max()), (float((gotx - refx).abs().max()), sx)
    refw = wr.grad
    sw = float(refw.abs().max())
    assert float((dw.cpu().double() - refw).abs().max()) <= TOL * sw, (float((dw.cpu().double() - refw).abs().max()), sw)
