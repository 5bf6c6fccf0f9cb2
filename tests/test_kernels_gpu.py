"""Parity of every C-ABI kernel (called through ctypes) against fp32 CPU references:
torch.nn.functional for the float ops, oracle/ for the losses and metrics."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = dict(atol=2e-4, rtol=2e-4)


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from miccai2021_cataract_semantic_segmentation_amd import ops as o
    return o


def dev(t):
    return t.cuda()


def nhwc(t):  # NCHW cpu -> NHWC gpu contiguous
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):  # NHWC gpu -> NCHW cpu
    return t.cpu().permute(0, 3, 1, 2)


def ohwi(w):  # [O,I,kh,kw] cpu -> channels_last gpu (physical OHWI)
    return w.cuda().contiguous(memory_format=torch.channels_last)


def close(a, b, atol=2e-4, rtol=2e-4):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-12
    assert err <= atol + rtol * scale, "max abs err %g (ref scale %g)" % (err, scale)


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, dil
    (2, 9, 11, 16, 32, 1, 1, 0, 1),
    (1, 12, 10, 64, 64, 3, 1, 1, 1),
    (2, 13, 17, 32, 48, 3, 2, 1, 1),
    (1, 20, 24, 16, 25, 1, 1, 0, 1),
    (1, 17, 19, 48, 160, 3, 1, 2, 2),
    (1, 30, 34, 16, 72, 3, 1, 12, 12),
    (2, 8, 8, 304, 20, 3, 1, 1, 1),
    (1, 16, 16, 128, 130, 1, 2, 0, 1),
    (3, 7, 5, 8, 8, 3, 1, 4, 4),
    # images smaller than one 16-row K step (UPerNet pyramid bins at tiny inputs)
    (2, 3, 4, 64, 64, 3, 1, 1, 1),
    (20, 1, 1, 32, 16, 1, 1, 0, 1),
    (5, 2, 3, 16, 16, 3, 1, 1, 1),
    (1, 2, 40, 8, 8, 3, 1, 1, 1),
    (3, 6, 6, 16, 12, 3, 2, 1, 1),
    # HRNet branch widths: 48 / 96-wide 16x16x4 tiles (fwd / dgrad) and the direct backward-weight kernel
    # (>= 1024 strips of 16 pixels; W not a multiple of 16 -> partial strips, halo at every border)
    (4, 64, 70, 48, 48, 3, 1, 1, 1),
    (2, 70, 120, 96, 96, 3, 1, 1, 1),
    (1, 33, 40, 48, 96, 3, 1, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(ops, case):
    B, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).requires_grad_()
    b = torch.randn(Cout, generator=g, requires_grad=True)
    y = F.conv2d(x, w, b, s, p, d)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, bd = nhwc(x.detach()), ohwi(w.detach()), b.detach().cuda()
    yd = ops.conv_fwd(xd, wd, bd, Cout, k, k, s, p, d)
    close(nchw(yd), y)
    # padded output with zero fill
    ldp = (Cout + 3) // 4 * 4 + 4
    yd2 = ops.conv_fwd(xd, wd, None, Cout, k, k, s, p, d, zero_to=ldp)
    assert ops.ld_of(yd2) == ldp
    close(nchw(yd2), F.conv2d(x, w, None, s, p, d))
    full = torch.as_strided(yd2, yd2.shape[:3] + (ldp,), yd2.stride())
    assert float(full[..., Cout:].abs().max()) == 0.0
    # backward: dy lives in a zero-padded buffer (ld multiple of 4)
    gyd = ops.new_act(B, y.shape[2], y.shape[3], Cout, xd.device, zero=True)
    gyd.copy_(nhwc(gy))
    dx = ops.conv_bwd_data(gyd, wd, tuple(xd.shape), k, k, s, p, d)
    close(nchw(dx), x.grad)
    dx2 = ops.conv_bwd_data(gyd, wd, tuple(xd.shape), k, k, s, p, d, out=dx.clone(), accumulate=True)
    close(nchw(dx2), 2 * x.grad)
    dw = torch.empty_like(wd)
    db = torch.empty(Cout, device="cuda")
    ops.conv_bwd_weight(xd, gyd, dw, db, k, k, s, p, d)
    close(dw.cpu(), w.grad, atol=5e-4, rtol=5e-4)
    close(db, b.grad, atol=5e-4, rtol=5e-4)


def test_conv_stem(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 37, 45, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, 2, 3)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    close(x4[..., :3].cpu(), x.permute(0, 2, 3, 1))
    assert float(x4[..., 3].abs().max()) == 0
    wd = ohwi(w.detach())
    pk = ops.stem_pack_weight(wd, 64)
    yd = ops.conv_fwd(x4, pk, None, 64, 7, 7, 2, 3, 1, stem4=True)
    close(nchw(yd), y)
    dpk = torch.empty_like(pk)
    ops.conv_bwd_weight(x4, nhwc(gy), dpk, None, 7, 7, 2, 3, 1, stem4=True)
    dw = torch.empty_like(wd)
    ops.stem_unpack_grad(dpk, dw, 64)
    close(dw.cpu(), w.grad, atol=1e-3, rtol=5e-4)


def test_conv_large_channels_splits(ops):
    """wide reduction (K = 9*256) and many pixel rows -> multi-split weight gradient"""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 34, 30, generator=g, requires_grad=True)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.02).requires_grad_()
    y = F.conv2d(x, w, None, 1, 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd = nhwc(x.detach()), ohwi(w.detach())
    close(nchw(ops.conv_fwd(xd, wd, None, 256, 3, 3, 1, 2, 2)), y)
    gyd = nhwc(gy)
    close(nchw(ops.conv_bwd_data(gyd, wd, tuple(xd.shape), 3, 3, 1, 2, 2)), x.grad)
    dw = torch.empty_like(wd)
    ops.conv_bwd_weight(xd, gyd, dw, None, 3, 3, 1, 2, 2)
    close(dw.cpu(), w.grad, atol=2e-3, rtol=5e-4)


def test_conv_concat_views(ops):
    """input read from / output written into channel slices of wider buffers (ld > C)"""
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 32, 6, 7, generator=g)
    w = torch.randn(16, 32, 1, 1, generator=g) * 0.2
    big_in = torch.zeros(1, 6, 7, 80, device="cuda")
    big_in[..., 48:80] = nhwc(x)
    big_out = torch.full((1, 6, 7, 40), 7.0, device="cuda")
    ops.conv_fwd(big_in[..., 48:80], ohwi(w), None, 16, 1, 1, out=big_out[..., 8:24])
    close(nchw(big_out[..., 8:24]), F.conv2d(x, w))
    assert float((big_out[..., :8] - 7).abs().max()) == 0 and float((big_out[..., 24:] - 7).abs().max()) == 0


@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_gemm_batched(ops, layout):
    g = torch.Generator().manual_seed(5)
    Bz, M, N, K = 3, 70, 25, 40
    if layout == "NT":
        A, Bm = torch.randn(Bz, M, K, generator=g), torch.randn(Bz, N, K, generator=g)
        ref = A @ Bm.transpose(1, 2)
        Ad, Bd = A.cuda(), Bm.cuda()
        lda, ldb = K, K
    elif layout == "NN":
        K = 25  # reduction over a zero-padded 25 -> 28/32 wide operand
        A, Bm = torch.randn(Bz, M, K, generator=g), torch.randn(Bz, K, 36, generator=g)
        N = 36
        ref = A @ Bm
        Ad = torch.zeros(Bz, M, 32, device="cuda")
        Ad[..., :K] = A.cuda()
        Bd = Bm.cuda()
        lda, ldb = 32, N
    else:
        M, N, K = 25, 68, 150
        A, Bm = torch.randn(Bz, K, M, generator=g), torch.randn(Bz, K, N, generator=g)
        ref = A.transpose(1, 2) @ Bm
        Ad = torch.zeros(Bz, K, 32, device="cuda")
        Ad[..., :M] = A.cuda()
        Bd = Bm.cuda()
        lda, ldb = 32, N
    ldc = (N + 3) // 4 * 4 + 4
    Cd = torch.full((Bz, M, ldc), 3.0, device="cuda")
    code = {"NT": ops.NT, "NN": ops.NN, "TN": ops.TN}[layout]
    ops.gemm(code, Bz, M, N, K, Ad, lda, Ad.stride(0), Bd, ldb, Bd.stride(0), Cd, ldc, Cd.stride(0), zero_to=ldc)
    close(Cd[..., :N], ref)
    assert float(Cd[..., N:].abs().max()) == 0
    ops.gemm(code, Bz, M, N, K, Ad, lda, Ad.stride(0), Bd, ldb, Bd.stride(0), Cd, ldc, Cd.stride(0), accumulate=True)
    close(Cd[..., :N], 2 * ref)


@pytest.mark.parametrize("shape,relu,res", [((2, 9, 11, 64), True, False), ((3, 5, 7, 48), True, True),
                                            ((2, 25, 1, 256), False, False), ((1, 40, 50, 12), True, True)])
def test_batchnorm(ops, shape, relu, res):
    B, H, W, C = shape
    g = torch.Generator().manual_seed(C)
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 3).requires_grad_()
    r = torch.randn(B, C, H, W, generator=g).requires_grad_() if res else None
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_()
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_()
    rm, rv = 0.1 * torch.randn(C, generator=g), 1 + 0.1 * torch.rand(C, generator=g)
    rm0, rv0 = rm.clone(), rv.clone()
    z = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        z = z + r
    if relu:
        z = F.relu(z)
    gz = torch.randn(z.shape, generator=g)
    z.backward(gz)
    yd = nhwc(y.detach())
    rmd, rvd = rm0.cuda(), rv0.cuda()
    stats, scale = ops.bn_train_stats(yd, gamma.detach().cuda(), 1e-5, 0.1, rmd, rvd)
    close(rmd, rm, atol=1e-5)
    close(rvd, rv, atol=1e-5)
    rd = nhwc(r.detach()) if res else None
    zd = ops.bn_apply(yd, stats[:C], scale, beta.detach().cuda(), rd, relu)
    close(nchw(zd), z, atol=1e-4)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dres = torch.empty_like(yd) if res else None
    dy = ops.bn_backward(nhwc(gz), zd, yd, stats, gamma.detach().cuda(), relu, dg, db, dres)
    close(nchw(dy), y.grad, atol=2e-4)
    close(dg, gamma.grad, atol=1e-3, rtol=1e-3)
    close(db, beta.grad, atol=1e-3, rtol=1e-3)
    if res:
        close(nchw(dres), r.grad)
    elif relu:
        # without a residual branch the ReLU mask can be recomputed from y instead of read from z: identical bits
        dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dy2 = ops.bn_backward(nhwc(gz), None, yd, stats, gamma.detach().cuda(), relu, dg2, db2, None, beta=beta.detach().cuda())
        assert torch.equal(dy2, dy) and torch.equal(dg2, dg) and torch.equal(db2, db)
    # eval mode
    ze = F.batch_norm(y.detach(), rm, rv, gamma.detach(), beta.detach(), False, 0.1, 1e-5)
    sc = ops.bn_eval_scale(gamma.detach().cuda(), rv.cuda(), 1e-5)
    close(nchw(ops.bn_apply(yd, rm.cuda(), sc, beta.detach().cuda(), None, False)), ze, atol=1e-4)


def test_batchnorm_many_rows(ops):
    g = torch.Generator().manual_seed(1)
    y = torch.randn(4, 64, 136, 120, generator=g) * 0.5 + 10.0  # large mean / std ratio
    gamma, beta = torch.ones(64), torch.zeros(64)
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    yd = nhwc(y)
    stats, scale = ops.bn_train_stats(yd, gamma.cuda(), 1e-5, 0.1, None, None)
    close(stats[:64], y.mean((0, 2, 3)), atol=1e-5)
    close(nchw(ops.bn_apply(yd, stats[:64], scale, beta.cuda(), None, False)), z, atol=2e-4)


def test_maxpool(ops):
    g = torch.Generator().manual_seed(2)
    x = F.relu(torch.randn(2, 8, 13, 18, generator=g)).requires_grad_()  # many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    yd, idx = ops.maxpool_fwd(nhwc(x.detach()))
    close(nchw(yd), y, atol=0, rtol=0)
    close(nchw(ops.maxpool_bwd(nhwc(gy), idx, (2, 13, 18, 8))), x.grad, atol=1e-6)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("sizes", [((6, 10), (48, 80)), ((17, 30), (68, 120)), ((5, 7), (5, 7)), ((9, 9), (20, 31))])
def test_bilinear(ops, align, sizes):
    (H, W), (Ho, Wo) = sizes
    g = torch.Generator().manual_seed(H * W)
    C = 25
    x = torch.randn(2, C, H, W, generator=g, requires_grad=True)
    y = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=align)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = ops.new_act(2, H, W, C, torch.device("cuda"), ld=32, zero=True)
    xd.copy_(nhwc(x.detach()))
    yd = ops.bilinear_fwd(xd, Ho, Wo, align)
    close(nchw(yd), y, atol=1e-5)
    dx = ops.bilinear_bwd(nhwc(gy), (2, H, W, C), align, zero_to=32)
    close(nchw(dx), x.grad, atol=1e-4)
    full = torch.as_strided(dx, dx.shape[:3] + (32,), dx.stride())
    assert float(full[..., C:].abs().max()) == 0


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("sizes", [((17, 30), (68, 120)), ((34, 60), (136, 240)), ((6, 10), (48, 80))])
def test_bilinear_forward_with_staged_rows_is_bit_identical(ops, align, sizes):
    """bilinear_fwd_lds_kernel (contiguous K-class output rows, source rows with a 16-byte-granular stride staged in LDS: the logits upsample of
    models/OCR.py:128-131) against bilinear_fwd_kernel<1> on the same values from a densely packed source (row stride 25: not 16-byte
    granular, so the launcher keeps the gather kernel): the same expression on the same operands -- bit for bit, also accumulating."""
    (H, W), (Ho, Wo) = sizes
    g = torch.Generator().manual_seed(H + W + Ho)
    C = 25
    x = torch.randn(2, H, W, C, generator=g).cuda()
    xp = ops.new_act(2, H, W, C, torch.device("cuda"), ld=32, zero=True)
    xp.copy_(x)
    ya = ops.bilinear_fwd(xp, Ho, Wo, align)            # padded rows: the LDS kernel
    yb = ops.bilinear_fwd(x.contiguous(), Ho, Wo, align)   # dense rows: the gather kernel
    assert ya.shape == yb.shape and torch.equal(ya, yb)
    base = torch.randn(ya.shape, generator=g).cuda()
    oa, ob = base.clone(), base.clone()
    ops.bilinear_fwd(xp, Ho, Wo, align, out=oa, accumulate=True)
    ops.bilinear_fwd(x.contiguous(), Ho, Wo, align, out=ob, accumulate=True)
    assert torch.equal(oa, ob)
    ref = F.interpolate(x.permute(0, 3, 1, 2).cpu(), size=(Ho, Wo), mode="bilinear", align_corners=align)
    close(nchw(ya), ref, atol=1e-5)


@pytest.mark.parametrize("case", [
    # (B, H, W, C, Ho, Wo, ld of dy, align, zero_to, accumulate): contiguous logits rows (mode 1), feature maps / channel slices of a wider
    # buffer (mode 2, several channel chunks, several column segments), the scalar fallback (C % 4 != 0 with a padded row stride)
    (2, 34, 60, 25, 136, 240, 25, False, 32, False), (2, 34, 60, 25, 136, 240, 25, True, 32, True), (1, 136, 240, 25, 544, 960, 25, False, 32, False),
    (2, 17, 30, 384, 136, 240, 720, False, 0, False), (2, 34, 60, 192, 136, 240, 192, False, 0, True), (2, 68, 120, 96, 136, 240, 720, False, 0, False),
    (3, 5, 7, 48, 5, 7, 48, False, 0, False), (2, 9, 9, 8, 20, 31, 8, True, 0, False), (1, 1, 1, 4, 7, 9, 4, False, 0, False),
    (2, 6, 10, 25, 48, 80, 32, False, 32, False), (2, 20, 31, 12, 9, 9, 12, False, 0, False)])
def test_bilinear_backward_in_one_launch_is_bit_identical_to_the_two_passes(ops, case):
    """catseg_bilinear_bwd as ONE launch with the intermediate row in LDS (bilinear_bwd_fused_kernel) against the two separable passes it
    replaces (catseg_debug_set_bilinear_bwd_fused(0)): the same sums in the same order -> bit-identical gradients, padding columns and
    accumulation included; and against autograd of F.interpolate (reference call sites: models/OCR.py:128-131, models/HRNetv2.py:505-508)"""
    B, H, W, C, Ho, Wo, ld, align, zero_to, acc = case
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(H * W + C)
    gy = torch.randn(B, Ho, Wo, C, generator=g)
    buf = torch.randn(B, Ho, Wo, ld, generator=g).to(dev)          # dy as a channel slice of a wider buffer
    buf[..., :C] = gy.to(dev)
    dy = buf[..., :C] if ld != C else buf
    base = torch.randn(B, H, W, max(zero_to, (C + 3) // 4 * 4), generator=g).to(dev)
    res = []
    for fused in (1, 0):
        ops.lib.catseg_debug_set_bilinear_bwd_fused(fused)
        out = base.clone()[..., :C] if acc else None
        if acc:
            o = base.clone()
            out = torch.as_strided(o, (B, H, W, C), o.stride())
        dx = ops.bilinear_bwd(dy, (B, H, W, C), align, out=out, zero_to=zero_to, accumulate=acc)
        torch.cuda.synchronize()
        full = torch.as_strided(dx, dx.shape[:3] + (ops.ld_of(dx),), dx.stride())
        res.append(full.clone())
    ops.lib.catseg_debug_set_bilinear_bwd_fused(1)
    assert torch.equal(res[0], res[1])
    x = torch.zeros(B, C, H, W, requires_grad=True)
    F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=align).backward(gy.permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)
    got = res[0][..., :C].cpu() - (base[..., :C].cpu() if acc else 0)
    assert float((got - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    if zero_to > C and not acc:
        assert float(res[0][..., C:zero_to].abs().max()) == 0


def test_global_avgpool(ops):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 72, 9, 13, generator=g)
    yd = ops.global_avgpool_fwd(nhwc(x))
    close(yd.reshape(2, 72), x.mean((2, 3)), atol=1e-6)
    dx = torch.ones(2, 9, 13, 72, device="cuda")
    gy = torch.randn(2, 72, generator=g)
    ops.global_avgpool_bwd(gy.cuda().reshape(2, 1, 1, 72), dx, True)
    close(nchw(dx), 1 + gy[:, :, None, None].expand(2, 72, 9, 13) / (9 * 13), atol=1e-6)


def test_global_avgpool_vector_kernel(ops):
    """gap_fwd4_kernel (C % 4 == 0, HW >= 256: 16-byte loads, 64 row lanes, four load chains): ragged HW (tail loops), a channel count that
    leaves the last block partly empty, a channel SLICE of a wider tensor (ld > C), against the fp64 mean; run twice: deterministic"""
    g = torch.Generator().manual_seed(18)
    for (B, C, H, W, wide) in ((2, 136, 20, 33, 0), (1, 64, 16, 16, 0), (2, 72, 31, 17, 24), (1, 2048, 17, 30, 0)):
        full = torch.randn(B, H, W, C + wide, generator=g).cuda()
        x = full[..., :C]
        y1 = ops.global_avgpool_fwd(x)
        y2 = ops.global_avgpool_fwd(x)
        ref = x.double().mean((1, 2))
        assert torch.equal(y1, y2)
        assert float((y1.reshape(B, C).double() - ref).abs().max()) < 2e-7 * max(1.0, float(ref.abs().max())) + 3e-7, (B, C, H, W)


def test_softmaxes(ops):
    g = torch.Generator().manual_seed(6)
    B, N, K, ld = 2, 300, 25, 32
    x = (3 * torch.randn(B, N, K, generator=g)).requires_grad_()
    y = F.softmax(x, dim=1)  # over the N pixels, per (b, k)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = torch.zeros(B, N, ld, device="cuda")
    xd[..., :K] = x.detach().cuda()
    yd = ops.softmax_spatial_fwd(xd, K)
    close(yd[..., :K], y, atol=1e-6)
    assert float(yd[..., K:].abs().max()) == 0
    gyd = torch.zeros(B, N, ld, device="cuda")
    gyd[..., :K] = gy.cuda()
    dx = ops.softmax_spatial_bwd(yd, gyd, torch.empty_like(xd), K)
    close(dx[..., :K], x.grad, atol=1e-6)
    # rows
    x2 = torch.randn(B * N, K, generator=g, requires_grad=True)
    sc = 256 ** -0.5
    y2 = F.softmax(sc * x2, dim=-1)
    g2 = torch.randn(y2.shape, generator=g)
    y2.backward(g2)
    x2d = torch.zeros(B * N, ld, device="cuda")
    x2d[:, :K] = x2.detach().cuda()
    y2d = ops.softmax_rows_fwd(x2d, K, sc)
    close(y2d[:, :K], y2, atol=1e-6)
    assert float(y2d[:, K:].abs().max()) == 0
    g2d = torch.zeros(B * N, ld, device="cuda")
    g2d[:, :K] = g2.cuda()
    close(ops.softmax_rows_bwd(y2d, g2d, K, sc)[:, :K], x2.grad, atol=1e-6)


def _lovasz_case(ops, logits, labels, tol_loss=2e-6, check_grad=True):
    from oracle import losses as OL
    lg = logits.clone().requires_grad_()
    ref = OL.lovasz_softmax(lg, labels)
    ref.backward()
    B, K, H, W = logits.shape
    ld = logits.permute(0, 2, 3, 1).reshape(-1, K).contiguous().cuda()
    lb = labels.reshape(-1).cuda()
    dl = torch.empty_like(ld)
    loss = ops.lovasz_softmax(ld, lb, 1.0, dl)
    assert abs(float(loss) - float(ref)) < tol_loss + 1e-5 * abs(float(ref)), (float(loss), float(ref))
    if check_grad:
        gref = lg.grad.permute(0, 2, 3, 1).reshape(-1, K)
        close(dl, gref, atol=1e-7, rtol=1e-3)
    return float(loss)


def test_lovasz_golden(ops, golden):
    g = golden("losses")
    for tag in "abc":
        lg, lb = torch.from_numpy(g[tag + "_logits"]), torch.from_numpy(g[tag + "_labels"])
        v = _lovasz_case(ops, lg, lb)
        assert abs(v - float(g[tag + "_loss"])) < 2e-6
        K = lg.shape[1]
        ld = lg.permute(0, 2, 3, 1).reshape(-1, K).contiguous().cuda()
        dl = torch.empty_like(ld)
        ops.lovasz_softmax(ld, lb.reshape(-1).cuda(), 1.0, dl)
        close(dl, torch.from_numpy(g[tag + "_grad"]).permute(0, 2, 3, 1).reshape(-1, K), atol=1e-7, rtol=1e-3)


def test_lovasz_edge_cases(ops):
    g = torch.Generator().manual_seed(12)
    # all pixels ignore-labelled -> no class present -> loss 0, zero gradient
    lg = torch.randn(1, 25, 8, 8, generator=g)
    ld = lg.permute(0, 2, 3, 1).reshape(-1, 25).contiguous().cuda()
    dl = torch.full_like(ld, 5.0)
    loss = ops.lovasz_softmax(ld, torch.full((64,), 25, dtype=torch.int64).cuda(), 1.0, dl)
    assert float(loss) == 0.0 and float(dl.abs().max()) == 0.0
    # single pixel, single class present
    _lovasz_case(ops, torch.randn(1, 8, 1, 1, generator=g), torch.tensor([[[3]]]))
    # ragged size (not a multiple of any tile), 17 classes, weight + accumulate
    lg = 2 * torch.randn(1, 17, 67, 131, generator=g)
    lb = torch.randint(0, 18, (1, 67, 131), generator=g)
    v = _lovasz_case(ops, lg, lb)
    ld = lg.permute(0, 2, 3, 1).reshape(-1, 17).contiguous().cuda()
    d1, d2 = torch.empty_like(ld), torch.zeros_like(ld)
    ops.lovasz_softmax(ld, lb.reshape(-1).cuda(), 1.0, d1)
    l2 = ops.lovasz_softmax(ld, lb.reshape(-1).cuda(), 0.4, d2, accumulate=True)
    assert abs(float(l2) - 0.4 * v) < 1e-6
    close(d2, 0.4 * d1, atol=1e-8, rtol=1e-5)


def test_lovasz_many_pixels_sorted_property(ops):
    """P = 2*544*480: size-independent checks against the float64 numpy oracle"""
    from oracle import losses as OL
    g = torch.Generator().manual_seed(13)
    lg = 2 * torch.randn(2, 25, 136, 120, generator=g)
    lg = F.interpolate(lg, size=(544, 480), mode="bilinear", align_corners=True)
    lb = torch.randint(0, 26, (2, 17, 15), generator=g)
    lb = lb.repeat_interleave(32, 1).repeat_interleave(32, 2)
    lb[lb == 7] = 2
    ref = OL.lovasz_softmax_np(lg.numpy(), lb.numpy())
    ld = lg.permute(0, 2, 3, 1).reshape(-1, 25).contiguous().cuda()
    dl = torch.empty_like(ld)
    loss = ops.lovasz_softmax(ld, lb.reshape(-1).cuda(), 1.0, dl)
    assert abs(float(loss) - ref) < 5e-6, (float(loss), ref)
    # gradient of a softmax-composed loss sums to zero over classes at every pixel
    assert float(dl.sum(1).abs().max()) < 1e-9 + 1e-4 * float(dl.abs().max())
    assert torch.isfinite(dl).all()


def test_cross_entropy(ops, golden):
    g = golden("losses")
    lg, lb = torch.from_numpy(g["ce_logits"]), torch.from_numpy(g["ce_labels"])
    ld = lg.permute(0, 2, 3, 1).reshape(-1, 17).contiguous().cuda()
    dl = torch.empty_like(ld)
    loss = ops.cross_entropy(ld, lb.reshape(-1).cuda(), 17, 1.0, dl)
    assert abs(float(loss) - float(g["ce_loss"])) < 1e-6
    close(dl, torch.from_numpy(g["ce_grad"]).permute(0, 2, 3, 1).reshape(-1, 17), atol=1e-8, rtol=1e-4)


def test_confusion_matrix(ops, golden):
    g = golden("metrics")
    for exp, K in ((1, 8), (2, 17), (3, 25)):
        lg, lb = torch.from_numpy(g["e%d_logits" % exp]), torch.from_numpy(g["e%d_labels" % exp])
        ld = lg.permute(0, 2, 3, 1).reshape(-1, K).contiguous().cuda()
        cm = ops.confusion_matrix(ld, lb.reshape(-1).cuda())
        assert np.array_equal(cm.cpu().numpy(), g["e%d_cm" % exp])


def test_adam(ops):
    from oracle import losses as OL
    g = torch.Generator().manual_seed(21)
    n = 1003
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.cuda(), m.cuda(), v.cuda()
    for step in (1, 2, 3):
        OL.adam_step(p, gr, m, v, step, 1e-3)
        ops.adam_step(pd, gr.cuda(), md, vd, 1e-3, step)
    close(pd, p, atol=1e-6)
    close(vd, v, atol=1e-7)


# ---------------------------------------------------------------------------------------------- split precision (bf16 x 3)
B3_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, dil
    (2, 21, 27, 64, 96, 3, 1, 1, 1), (1, 30, 34, 48, 40, 3, 1, 2, 2), (2, 19, 23, 720, 512, 3, 1, 1, 1), (2, 24, 24, 256, 256, 1, 1, 0, 1),
    (2, 33, 29, 96, 192, 3, 2, 1, 1), (1, 16, 20, 24, 25, 1, 1, 0, 1), (2, 40, 44, 16, 512, 3, 1, 1, 1), (1, 68, 120, 256, 256, 3, 1, 12, 12),
    (3, 9, 7, 8, 8, 3, 1, 1, 1),
]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("case", B3_CASES)
def test_conv_bf16x3_split_precision(ops, case, tile):
    """catseg_split3 + catseg_conv2d_fwd_bf16x3 / _bwd_data_bf16x3 (six bf16 MFMA partial products per block) against an fp64
    F.conv2d: the error must be of fp32 size (2e-5 of the output scale; measured 2e-7 ... 3e-6), for every block-tile form,
    on inputs with a wide dynamic range (per-channel scales e^(2 N(0,1)))"""
    from miccai2021_cataract_semantic_segmentation_amd import _lib
    B, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(2 * torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr = x.double().requires_grad_()
    y64 = F.conv2d(xr, w.double(), b.double(), s, p, d)
    gy = torch.randn(y64.shape, generator=g)
    y64.backward(gy.double())
    xd, wd = nhwc(x), ohwi(w)
    try:
        _lib.lib.catseg_debug_set_b3_tile(tile)
        ld = (Cout + 3) // 4 * 4 + 4
        y = ops.conv_fwd_b3(tuple(xd.shape), ops.split3(xd), ops.split3_weight(wd), b.cuda(), Cout, k, k, s, p, d, zero_to=ld)
        close(nchw(y), y64.detach(), atol=0, rtol=2e-5)
        full = torch.as_strided(y, y.shape[:3] + (ld,), y.stride())
        assert float(full[..., Cout:].abs().max()) == 0.0
        if s == 1:
            gyd = ops.new_act(B, y64.shape[2], y64.shape[3], Cout, xd.device, zero=True)
            gyd.copy_(nhwc(gy))
            dx = ops.conv_bwd_data_b3(ops.split3(gyd), ops.split3_weight_t(wd), tuple(xd.shape), Cout, k, k, s, p, d)
            close(nchw(dx), xr.grad, atol=0, rtol=2e-5)
            dx2 = ops.conv_bwd_data_b3(ops.split3(gyd), ops.split3_weight_t(wd), tuple(xd.shape), Cout, k, k, s, p, d, out=dx.clone(), accumulate=True)
            close(nchw(dx2), 2 * xr.grad, atol=0, rtol=2e-5)
    finally:
        _lib.lib.catseg_debug_set_b3_tile(0)


def test_split3_blocked_layout(ops):
    """catseg_split3_blocked: blocked planes [3][ceil(C/16)][rows][16] and, from the same pass, the planar planes of catseg_split3 --
    both bit-identical to the planar split (channel tail of the last chunk zero), for row / channel counts off the 64 x 128 block"""
    for (rows_shape, C, ld) in [((3, 7, 11), 720, 720), ((2, 5, 13), 40, 44), ((1, 9, 9), 16, 16), ((2, 33, 4), 200, 200), ((1, 1, 70), 136, 140)]:
        g = torch.Generator().manual_seed(C + ld)
        full = torch.randn(rows_shape + (ld,), generator=g).cuda() * 3.0
        x = full[..., :C] if ld != C else full
        ref = ops.split3(x)                                  # [3, rows, roundup(C, 8)]
        blk, planar = ops.split3_blocked(x, with_planar=True)
        assert torch.equal(planar, ref)
        rows = ref.shape[1]
        c16 = (C + 15) // 16
        want = torch.zeros((3, rows, c16 * 16), dtype=torch.int16, device="cuda")
        want[:, :, :ref.shape[2]] = ref
        want = want.view(3, rows, c16, 16).permute(0, 2, 1, 3).contiguous()
        assert torch.equal(blk, want)
        assert torch.equal(ops.split3_blocked(x)[0], want)
    w = torch.randn(72, 48, 3, 3).cuda().contiguous(memory_format=torch.channels_last)
    ref = ops.split3_weight(w)                               # [3, O, K]
    want = ref.view(3, 72, 27, 16).permute(0, 2, 1, 3).contiguous()
    assert torch.equal(ops.split3_weight_blocked(w), want)
    w = torch.randn(40, 24, 3, 3).cuda().contiguous(memory_format=torch.channels_last)
    ref = ops.split3_weight_t(w)                             # [3, Cin, taps, roundup(O, 8)]
    pad = torch.zeros((3, 24, 9, 48), dtype=torch.int16, device="cuda")
    pad[..., :40] = ref
    want = pad.view(3, 24, 27, 16).permute(0, 2, 1, 3).contiguous()
    assert torch.equal(ops.split3_weight_t_blocked(w), want)


B3_BLOCKED_CASES = [(2, 19, 23, 720, 512, 3, 1, 1, 1), (2, 24, 24, 256, 256, 1, 1, 0, 1), (1, 17, 21, 64, 200, 3, 1, 1, 1), (2, 33, 29, 96, 320, 3, 2, 1, 1),
                    (1, 40, 44, 16, 512, 3, 1, 1, 1), (1, 36, 40, 256, 256, 3, 1, 6, 6), (1, 16, 16, 1024, 512, 1, 1, 0, 1), (3, 9, 7, 32, 24, 3, 1, 1, 1)]


@pytest.mark.parametrize("case", B3_BLOCKED_CASES)
def test_conv_bf16x3_blocked_planes(ops, case):
    """forward / backward-data of the 256 x 256 kernel from BLOCKED planes against an fp64 F.conv2d (as test_conv_bf16x3_split_precision)"""
    B, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(2 * torch.randn(1, Cin, 1, 1, generator=g))
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr = x.double().requires_grad_()
    y64 = F.conv2d(xr, w.double(), b.double(), s, p, d)
    gy = torch.randn(y64.shape, generator=g)
    y64.backward(gy.double())
    xd, wd = nhwc(x), ohwi(w)
    ld = (Cout + 3) // 4 * 4 + 4
    y = ops.conv_fwd_b3_blocked(tuple(xd.shape), ops.split3_blocked(xd)[0], ops.split3_weight_blocked(wd), b.cuda(), Cout, k, k, s, p, d, zero_to=ld)
    close(nchw(y), y64.detach(), atol=0, rtol=2e-5)
    full = torch.as_strided(y, y.shape[:3] + (ld,), y.stride())
    assert float(full[..., Cout:].abs().max()) == 0.0
    if s == 1:
        gyd = ops.new_act(B, y64.shape[2], y64.shape[3], Cout, xd.device, zero=True)
        gyd.copy_(nhwc(gy))
        dyb, wtb = ops.split3_blocked(gyd)[0], ops.split3_weight_t_blocked(wd)
        dx = ops.conv_bwd_data_b3_blocked(dyb, wtb, tuple(xd.shape), Cout, k, k, s, p, d)
        close(nchw(dx), xr.grad, atol=0, rtol=2e-5)
        dx2 = ops.conv_bwd_data_b3_blocked(dyb, wtb, tuple(xd.shape), Cout, k, k, s, p, d, out=dx.clone(), accumulate=True)
        close(nchw(dx2), 2 * xr.grad, atol=0, rtol=2e-5)


@pytest.mark.parametrize("case", [(2, 19, 23, 64, 256, 3, 1, 1, 1, True, True), (1, 24, 24, 256, 320, 1, 1, 0, 1, False, True),
                                  (3, 17, 21, 32, 200, 3, 2, 1, 1, True, False), (2, 16, 20, 48, 208, 3, 1, 2, 2, False, False)])
def test_conv_fused_inference_bf16x3(ops, case):
    """inference epilogue of the 256 x 256 bf16x3 kernel (bias + residual + ReLU, blocked planes) through ops.conv_fwd_fused with the
    layer thresholds lowered, against fp64; the fp32 fused kernel must agree to fp32 rounding"""
    B, H, W, Cin, Cout, k, s, p, d, use_res, relu = case
    g = torch.Generator().manual_seed(sum(case[:9]) + 5)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g)
    y64 = F.conv2d(x.double(), w.double(), b.double(), s, p, d)
    res = torch.randn(y64.shape, generator=g) if use_res else None
    if use_res:
        y64 = y64 + res.double()
    if relu:
        y64 = y64.relu()
    xd, wd = nhwc(x), ohwi(w)
    resd = nhwc(res) if use_res else None
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES)
    try:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = "bf16x3", 1, 16, 16, 1
        prof, ops.PROFILE = ops.PROFILE, []
        y = ops.conv_fwd_fused(xd, wd, b.cuda(), resd, relu, Cout, k, k, s, p, d)
        kinds = [q[0] for q in ops.PROFILE]
        ops.PROFILE = prof
        assert "fwd_b3" in kinds or "fwd_h2" in kinds, kinds     # a split-precision kernel ran, not the fp32 one
        ops.PRECISION = "fp32"
        y32 = ops.conv_fwd_fused(xd, wd, b.cuda(), resd, relu, Cout, k, k, s, p, d)
    finally:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = saved
    close(nchw(y), y64, atol=0, rtol=2e-5)
    close(nchw(y32), y64, atol=0, rtol=2e-5)
    if B > 1:
        # the batch cut that keeps a piece's three blocked planes below the buffer-resource limit (4 GB; here: one image per piece)
        saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_PLANE_LIMIT)
        try:
            ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = "bf16x3", 1, 16, 16, 1
            ops.B3_PLANE_LIMIT = 6 * H * W * Cin + 1
            yc = ops.conv_fwd_fused(xd, wd, b.cuda(), resd, relu, Cout, k, k, s, p, d)
        finally:
            ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_PLANE_LIMIT = saved
        assert torch.equal(yc, y)


B3_WGRAD_CASES = B3_CASES + [(2, 70, 90, 64, 64, 3, 1, 1, 1), (1, 68, 120, 720, 512, 3, 1, 1, 1), (3, 40, 40, 128, 300, 3, 2, 1, 1)]


@pytest.mark.parametrize("case", B3_WGRAD_CASES)
def test_conv_bwd_weight_bf16x3(ops, case):
    """catseg_conv2d_bwd_weight_bf16x3 (pixel-major operands, ds_read_b64_tr_b16 transposed fragments, split reduction) and
    catseg_bias_grad against an fp64 F.conv2d weight gradient"""
    import ctypes
    from miccai2021_cataract_semantic_segmentation_amd import _lib
    B, H, W, Cin, Cout, k, s, p, d = case
    if Cin % 8:
        pytest.skip("Cin % 8 != 0 stays on the fp32 kernel")
    g = torch.Generator().manual_seed(sum(case) + 2)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(1, Cin, 1, 1, generator=g))
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).double().requires_grad_()
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, b, s, p, d)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    xd = nhwc(x)
    gyd = ops.new_act(B, y.shape[2], y.shape[3], Cout, xd.device, zero=True)
    gyd.copy_(nhwc(gy))
    dw = torch.empty((Cout, Cin, k, k), device="cuda").contiguous(memory_format=torch.channels_last)
    db = torch.empty(Cout, device="cuda")
    desc = ops.make_desc(xd.shape, Cin, Cout, (Cout + 7) // 8 * 8, k, k, s, p, d)
    ws = ops.workspace(_lib.lib.catseg_conv2d_bwd_weight_bf16x3_workspace(ctypes.byref(desc)) + 256 * Cout * 4, xd.device)
    xp, dyp = ops.split3(xd), ops.split3(gyd)      # (keep both alive: the planes are only referenced by raw pointers)
    _lib.check(_lib.lib.catseg_conv2d_bwd_weight_bf16x3(ctypes.byref(desc), ops.ptr(xp), ops.ptr(dyp), ops.ptr(dw),
                                                        ops.ptr(ws), ws.numel(), ops.stream()))
    _lib.check(_lib.lib.catseg_bias_grad(ops.ptr(gyd), ops.ld_of(gyd), ops.rows_of(gyd), Cout, ops.ptr(db), ops.ptr(ws), ws.numel(), ops.stream()))
    close(dw.cpu(), w.grad, atol=0, rtol=3e-5)
    close(db, b.grad, atol=0, rtol=3e-5)


def test_split3_is_exact(ops):
    """x == h + m + l exactly (to the last bit) over 30 orders of magnitude (values below ~1e-33, whose third piece would be a
    bf16 subnormal, are split with an absolute error < 1e-38: irrelevant), pad columns are zero"""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(1000, 20, generator=g) * torch.exp(8 * torch.randn(1000, 1, generator=g)).clamp(1e-15, 1e15)).cuda()
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, 3.0e-30])
    pl = ops.split3(x.view(1, 1000, 1, 20))
    assert pl.shape == (3, 1000, 24)
    parts = pl.view(torch.bfloat16).float()
    assert float(parts[:, :, 20:].abs().max()) == 0.0
    s = (parts[0, :, :20].double() + parts[1, :, :20].double() + parts[2, :, :20].double()).float()
    assert torch.equal(s, x)


BNSTAT_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad  (fp32 tile forms: 64x64 ... 256x128 and the 48 / 96-wide 16x16x4 forms; bias on)
    (2, 21, 27, 64, 64, 3, 1, 1), (2, 33, 41, 64, 256, 1, 1, 0), (1, 70, 90, 256, 512, 3, 1, 1), (4, 64, 70, 48, 48, 3, 1, 1),
    (2, 70, 120, 96, 96, 3, 1, 1), (2, 9, 7, 32, 40, 3, 2, 1), (1, 5, 3, 16, 16, 1, 1, 0), (2, 40, 44, 128, 192, 3, 1, 1),
]


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", BNSTAT_CASES)
def test_conv_epilogue_bn_statistics(ops, case, mode):
    """BatchNorm batch statistics from the convolution epilogue (catseg_conv2d_fwd_bnstats / _bf16x3_bnstats + catseg_bn_finalize)
    against torch's float64 statistics of the convolution output, and against the separate statistics pass (catseg_bn_train_stats)"""
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g) + 0.5
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 3          # a large bias: the shifted sums must cope with mean >> std
    y64 = F.conv2d(x.double(), w.double(), b.double(), s, p).permute(0, 2, 3, 1).reshape(-1, Cout)
    xd, wd = nhwc(x), ohwi(w)
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES)
    ops.PRECISION = mode
    if mode == "bf16x3":
        ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = 1, 16, 16, 1
    try:
        y, partials = ops.conv_fwd(xd, wd, b.cuda(), Cout, k, k, s, p, 1, bn_stats=True)
    finally:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = saved
        ops.release_b3_cache()
    if partials is None:        # 16 / 32-wide tile forms (class logits, grouped convolutions) have no fused statistics
        assert mode == "fp32" and Cout <= 32
        return
    gamma = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda()
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    stats, scale = ops.bn_finalize(partials, y64.shape[0], Cout, gamma, 1e-5, 0.1, rm, rv)
    mean64, var64 = y64.mean(0), y64.var(0, unbiased=False)
    close(stats[:Cout], mean64, atol=1e-6, rtol=2e-6)
    close(stats[Cout:], 1.0 / torch.sqrt(var64 + 1e-5), atol=0, rtol=2e-5)
    close(scale, gamma.cpu().double() / torch.sqrt(var64 + 1e-5), atol=0, rtol=2e-5)
    n = y64.shape[0]
    close(rm, 0.1 * mean64, atol=1e-6, rtol=1e-5)
    close(rv, 0.9 + 0.1 * var64 * n / max(n - 1, 1), atol=0, rtol=2e-5)
    # the two-kernel path gives the same statistics to fp32 rounding
    rm2, rv2 = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    stats2, _ = ops.bn_train_stats(y, gamma, 1e-5, 0.1, rm2, rv2)
    close(stats, stats2.cpu(), atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize("name", ["per_image", "ignore25", "all", "list", "per_image_ignore_list_e2"])
def test_lovasz_options_match_reference(golden, name):
    """per_image / classes_to_ignore / classes_to_consider of the reference's LovaszSoftmax (fixtures from the REAL reference)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from miccai2021_cataract_semantic_segmentation_amd.losses import LovaszSoftmax
    cfgs = {"per_image": {"experiment": 3, "per_image": True}, "ignore25": {"experiment": 3, "classes_to_ignore": 25},
            "all": {"experiment": 3, "classes_to_consider": "all"}, "list": {"experiment": 3, "classes_to_consider": [0, 3, 7, 12, 24]},
            "per_image_ignore_list_e2": {"experiment": 2, "per_image": True, "classes_to_ignore": 17, "classes_to_consider": [1, 2, 5, 16]}}
    g = golden("lovasz_options")
    lg = torch.from_numpy(g[name + ":logits"]).cuda().requires_grad_()
    lb = torch.from_numpy(g[name + ":labels"]).cuda()
    loss = LovaszSoftmax(dict(cfgs[name]))(lg, lb)
    loss.backward()
    assert abs(float(loss) - float(g[name + ":loss"])) < 2e-6, (float(loss), float(g[name + ":loss"]))
    close(lg.grad, torch.from_numpy(g[name + ":grad"]), atol=1e-7, rtol=1e-3)


@pytest.mark.parametrize("case", [(2, 12, 14, 64, 64, 3, 1, 1, 1, 8), (1, 17, 19, 128, 256, 3, 2, 1, 1, 32), (2, 9, 9, 32, 32, 3, 1, 2, 2, 4)])
def test_grouped_conv_fwd_bwd(ops, case):
    """grouped convolution (torchvision ResNeXt, models/ResNeXt.py:29-60): forward, backward-data and backward-weight vs F.conv2d(groups)"""
    B, H, W, Cin, Cout, k, s, p, d, G = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin // G, k, k, generator=g) * 0.1).requires_grad_()
    y = F.conv2d(x, w, None, s, p, d, G)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, gyd = nhwc(x.detach()), ohwi(w.detach()), nhwc(gy)
    close(nchw(ops.conv_fwd(xd, wd, None, Cout, k, k, s, p, d, groups=G)), y)
    close(nchw(ops.conv_bwd_data(gyd, wd, tuple(xd.shape), k, k, s, p, d, groups=G)), x.grad)
    dw = torch.empty_like(wd)
    ops.conv_bwd_weight(xd, gyd, dw, None, k, k, s, p, d, groups=G)
    close(dw.cpu(), w.grad, atol=5e-4, rtol=5e-4)


@pytest.mark.parametrize("case", [(2, 43, 55, 48, 96, 3, 2, 1, 1), (1, 40, 64, 96, 192, 3, 2, 1, 1), (2, 21, 20, 64, 40, 3, 2, 1, 1),
                                  (1, 33, 31, 32, 48, 1, 2, 0, 1), (1, 37, 41, 24, 64, 3, 2, 2, 2), (3, 5, 7, 256, 64, 3, 2, 1, 1)])
@pytest.mark.parametrize("multi", [1, 0])
def test_strided_backward_data_parity_classes(ops, case, multi):
    """stride-2 backward-data: the four input-pixel parity classes in ONE launch (igemm_f32_multi_kernel; odd extents give the
    classes different row counts, 1x1 / dilated filters leave classes without a tap) and as four launches, vs torch"""
    from miccai2021_cataract_semantic_segmentation_amd import _lib
    B, H, W, Ci, Co, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    w = torch.randn(Co, Ci, k, k, generator=g) * (2.0 / (Ci * k * k)) ** 0.5
    y = F.conv2d(x, w, None, s, p, d)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)
    gyd = ops.new_act(B, y.shape[2], y.shape[3], Co, torch.device("cuda"), zero=True)
    gyd.copy_(gy.permute(0, 2, 3, 1))
    try:
        _lib.check(_lib.lib.catseg_debug_set_strided_multi(multi))
        dx = ops.conv_bwd_data(gyd, wd, (B, H, W, Ci), k, k, s, p, d)
        base = torch.full((B, H, W, Ci), 0.5, device="cuda")
        acc = ops.conv_bwd_data(gyd, wd, (B, H, W, Ci), k, k, s, p, d, out=base.clone(), accumulate=True)
    finally:
        _lib.lib.catseg_debug_set_strided_multi(1)
    ref = x.grad.permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    assert float((dx.cpu() - ref).abs().max()) <= 2e-5 * sc + 2e-6
    assert float((acc.cpu() - ref - 0.5).abs().max()) <= 2e-5 * sc + 2e-6


def test_fused_inference_convs_share_one_split_of_their_input(ops):
    """two fused bf16x3 inference convolutions that read the same tensor (ASPP branches, the two OCR head convolutions, UPerNet)
    split it once: the plane cache is keyed by the tensor's identity (advisor finding, round 2)"""
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES)
    calls = []
    real, real2 = ops.split3_blocked, ops.split2h
    try:
        ops.PRECISION = "bf16x3"
        ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = 1, 64, 32, 1
        ops.split3_blocked = lambda x, with_planar=False: (calls.append(1), real(x, with_planar))[1]
        ops.split2h = lambda x, blocked=True, planar=False: (calls.append(1), real2(x, blocked, planar))[1]     # (CATSEG_HEADS=f16x2, the default)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(2, 12, 20, 64, generator=g).cuda()
        outs = []
        for seed in (1, 2):
            w = (torch.randn(208, 64, 3, 3, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last)
            outs.append((w, ops.conv_fwd_fused(x, w, None, None, True, 208, 3, 3, 1, 1, 1)))
        assert len(calls) == 1, "the shared input was split %d times" % len(calls)
        for w, y in outs:
            ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1, 1)).permute(0, 2, 3, 1)
            close(y, ref, atol=0, rtol=2e-5)
    finally:
        ops.split3_blocked, ops.split2h = real, real2
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES = saved
        ops.release_b3_cache()
