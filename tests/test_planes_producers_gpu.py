"""The producers of fp16 x 2 operand planes (csrc/norm.hip: catseg_bn_finalize_counts_bound, catseg_bn_apply_planes,
catseg_bn_backward_planes / _pre_planes; csrc/pointwise.hip: catseg_add_n_act_planes) through the C ABI, against the fp32 kernels they
extend: same fp32 results bit for bit, planes that reproduce them to 2^-22, exponents that can never overflow fp16 (they come from
bounds proven before the pass), and the true max| | left in the records."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _rebuild(pl):
    """fp32 tensor [B, H, W, C] from planes: (h + l) * 2^-e"""
    B, H, W, C = pl.shape
    hl = pl.buf.view(torch.float16).view(2, C // 8, B * H * W, 8).float()
    assert torch.isfinite(hl).all()
    return (hl[0] + hl[1]).permute(1, 0, 2).reshape(B, H, W, C) * 2.0 ** (-int(pl.rec[1]))


def _close_to_planes(pl, ref):
    e = int(pl.rec[1])
    d = (_rebuild(pl) - ref).abs()
    assert float((d - ref.abs() * 2.0 ** -22).max()) <= 2.0 ** (-25 - e), float(d.max())
    bound = pl.rec[3:4].view(torch.float32)          # (word 3: the bound the exponent was derived from)
    assert float(ref.abs().max()) <= float(bound) * (1 + 1e-6), (float(ref.abs().max()), float(bound))
    assert float(bound) * 2.0 ** e < 2.0 ** 15 * 1.002                       # planes in range by construction


@pytest.mark.parametrize("C,shape", [(48, (2, 19, 37)), (96, (2, 9, 33)), (192, (1, 7, 45)), (384, (2, 5, 30))])
@pytest.mark.parametrize("residual,relu", [(False, True), (True, True), (False, False)])
def test_conv_bn_planes_chain_forward_and_backward(C, shape, residual, relu):
    """conv (planes in, max|y| out) -> finalize (+ bound) -> apply (+ planes) and its backward (dy as planes only) against the fp32 kernels"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    B, H, W = shape
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H, W, C, generator=g) * torch.exp(torch.randn(C, generator=g))).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.3).to(dev)
    res = None
    if residual:
        res = (torch.randn(B, H, W, C, generator=g) * 2).to(dev)
        res._amax = ops.new_amax(dev)
        res._amax[64:65] = res.abs().max().reshape(1).view(torch.int32)
    rows = B * H * W
    saved = (ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS)
    ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS = "f16x2", "bf16x3", True, 1
    try:
        xp = ops.planes_from_f32(x)
        yrec = ops.new_amax(dev)
        y, partials = ops.dconv3_pl(xp, ops.dconv3_weight_image(w, h2=True), None, bn_stats=True, out_rec=yrec)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        zrec = ops.new_amax(dev)
        stats, scale = ops.bn_finalize(partials, rows, C, gamma, 1e-5, 0.1, rm, rv, bound=(beta, yrec, zrec))
        stats2, scale2 = ops.bn_finalize(partials, rows, C, gamma, 1e-5, 0.1, rm.clone(), rv.clone())
        assert torch.equal(stats, stats2) and torch.equal(scale, scale2)
        z_ref = ops.bn_apply(y, stats[:C], scale, beta, res, relu)
        z = ops.bn_apply(y, stats[:C], scale, beta, res, relu, planes_rec=zrec)
        assert torch.equal(z, z_ref)
        _close_to_planes(z._planes, z_ref)
        assert float(zrec.view(torch.float32)[::32].max()) == float(z_ref.abs().max())        # the true max|z| in the amax slots
        if not residual and relu:
            zrec2 = ops.new_amax(dev)
            ops.bn_finalize(partials, rows, C, gamma, 1e-5, 0.1, rm, rv, bound=(beta, yrec, zrec2))
            zo = ops.bn_apply(y, stats[:C], scale, beta, None, True, out=torch.full_like(y, float("nan")), planes_rec=zrec2, planes_only=True)
            assert torch.isnan(zo).all() and zo._planes_only                                  # the fp32 tensor is never written
            _close_to_planes(zo._planes, z_ref)
        # ---- backward: dy as planes only
        dz = (torch.randn(B, H, W, C, generator=g) * 1e-3).to(dev)
        z_mask = z_ref if (residual or not relu) else None
        dg1, db1, dg2, db2 = (torch.empty(C, device=dev) for _ in range(4))
        dres1 = torch.zeros_like(y) if residual else None
        dres2 = torch.zeros_like(y) if residual else None
        dy_ref = ops.bn_backward(dz, z_mask, y, stats, gamma, relu, dg1, db1, dres1, False, beta=beta)
        dyp = ops.bn_backward_planes(dz, z_mask, y, stats, gamma, relu, dg2, db2, dres2, False, beta, yrec)
        assert torch.equal(dg1, dg2) and torch.equal(db1, db2)
        if residual:
            assert torch.equal(dres1, dres2)
        _close_to_planes(dyp, dy_ref)
        # ... and the two kernels that consume them, against the in-kernel-split kernels fed the fp32 dy
        dw1, dw2 = torch.empty_like(w), torch.empty_like(w)
        ops.dwgrad3_pl(xp, dyp, dw1)
        x._amax, dy_ref._amax = xp.rec, None
        ops.PLANES = False
        ops.dwgrad3(x, dy_ref, dw2)          # (dy without a record: the three-plane bf16 kernel, exact operands)
        ops.PLANES = True
        assert float((dw1 - dw2).abs().max()) <= 2e-5 * float(dw2.abs().max())
        dx1 = ops.conv_bwd_data_pl(dyp, w, torch.full_like(x, float("nan")))
        dx2 = ops.dconv3(dy_ref, ops.dconv3_weight_image(w, backward_data=True))
        assert float((dx1 - dx2).abs().max()) <= 2e-5 * float(dx2.abs().max())
    finally:
        ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS = saved
        ops.release_b3_cache()


def test_add_n_act_planes_and_bilinear_record_propagation():
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    saved = (ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS)
    ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS = "f16x2", "bf16x3", True, 1
    try:
        small = torch.randn(2, 5, 8, 96, generator=g).to(dev)
        small._amax = ops.new_amax(dev)
        small._amax[0:1] = small.abs().max().reshape(1).view(torch.int32)
        up = ops.bilinear_fwd(small, 10, 16, False)
        assert ops.amax_of(up) is small._amax and float(up.abs().max()) <= float(small.abs().max())
        terms = [up]
        for i in range(2):
            t = (torch.randn(2, 10, 16, 96, generator=g) * (i + 1)).to(dev)
            t._amax = ops.new_amax(dev)
            t._amax[32:33] = t.abs().max().reshape(1).view(torch.int32)
            terms.append(t)
        ops.PLANES = False
        ref = ops.add_n_act(terms, True)
        ops.PLANES = True
        out = ops.add_n_act(terms, True)
        assert torch.equal(out, ref) and ops.planes_of(out) is not None
        _close_to_planes(out._planes, ref)
        assert float(out._amax.view(torch.float32)[::32].max()) == float(ref.abs().max())
        terms[1]._amax = None              # a term without a record: no bound, no planes
        assert ops.planes_of(ops.add_n_act(terms, True)) is None
    finally:
        ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS = saved
        ops.release_b3_cache()


def test_backward_data_with_fused_bn_backward_pass_on_planes():
    """catseg_dconv3_pl_bnbwd + catseg_bn_backward_pre_planes against the two-pass route (dx by the plain kernel, then catseg_bn_backward)"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    C, B, H, W = 48, 2, 19, 37
    saved = (ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS)
    ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS = "f16x2", "bf16x3", True, 1
    try:
        x0 = torch.randn(B, H, W, C, generator=g).to(dev)
        w1 = (torch.randn(C, C, 3, 3, generator=g) * 0.07).to(dev).contiguous(memory_format=torch.channels_last)
        w2 = (torch.randn(C, C, 3, 3, generator=g) * 0.07).to(dev).contiguous(memory_format=torch.channels_last)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
        rows = B * H * W
        xp = ops.planes_from_f32(x0)
        yrec = ops.new_amax(dev)
        q, partials = ops.dconv3_pl(xp, ops.dconv3_weight_image(w1, h2=True), None, bn_stats=True, out_rec=yrec)       # conv1
        zrec = ops.new_amax(dev)
        stats, scale = ops.bn_finalize(partials, rows, C, gamma, 1e-5, 0.1, torch.zeros(C, device=dev), torch.ones(C, device=dev), bound=(beta, yrec, zrec))
        dy2 = (torch.randn(B, H, W, C, generator=g) * 1e-3).to(dev)          # gradient of conv2's output
        dy2p = ops.planes_from_f32(dy2)
        # two-pass reference: dz by the plain kernel, then the full BatchNorm backward (mask recomputed from q)
        dz = ops.conv_bwd_data_pl(dy2p, w2, torch.empty_like(x0))
        dg1, db1, dg2, db2 = (torch.empty(C, device=dev) for _ in range(4))
        dq_ref = ops.bn_backward(dz, None, q, stats, gamma, True, dg1, db1, None, False, beta=beta)
        # fused: masked g + per-wave sums + max|g| from the epilogue, merge + apply writing planes
        gbuf, pre = ops.conv_bwd_data_pl(dy2p, w2, torch.full_like(x0, float("nan")), bn_src=(q, stats, gamma, beta))
        assert len(pre) == 3 and float(pre[2].view(torch.float32)[::32].max()) == float(gbuf.abs().max())
        dqp = ops.bn_backward_pre_planes(gbuf, q, stats, gamma, pre, yrec, dg2, db2)
        assert float((dg1 - dg2).abs().max()) <= 2e-5 * float(dg1.abs().max()) and float((db1 - db2).abs().max()) <= 2e-5 * float(db1.abs().max())
        d = (_rebuild(dqp) - dq_ref).abs()
        assert float(d.max()) <= 2e-5 * float(dq_ref.abs().max())
    finally:
        ops.TRUNK, ops.PRECISION, ops.PLANES, ops.DCONV3_MIN_ROWS = saved
        ops.release_b3_cache()
