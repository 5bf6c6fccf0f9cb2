"""The K-class classifier fused with the BatchNorm + ReLU in front of it (csrc/headfuse.h, ops.head_fwd / head_backward, engine._fused_head;
the layers are models/OCR.py:72-74, 97 of the reference: interm_prediction_head[1..4], conv_bn_dropout[1..2] + conv_out): the kernels against
an fp64 evaluation of the three separate layers and their autograd, and the layer through the engine against the separate-pass route."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ops():
    from miccai2021_cataract_semantic_segmentation_amd import ops as o
    yield o
    o.release_b3_cache()


def _planes_to_f64(blk, scale, C):
    e = int(scale.cpu()[1])
    hl = blk.cpu().view(torch.float16).double()          # [2, C/16, rows, 16]
    v = (hl[0] + hl[1]) * 2.0 ** -e
    return v.permute(1, 0, 2).reshape(v.shape[1], -1)[:, :C], e        # [rows, C]


@pytest.mark.parametrize("shape", [(2, 17, 23, 512, 25), (1, 37, 19, 64, 7), (3, 16, 16, 256, 32), (1, 8, 4, 128, 25), (2, 40, 64, 512, 25)])
def test_kernels_vs_fp64(ops, shape):
    B, H, W, C, K = shape
    rows = B * H * W
    g = torch.Generator().manual_seed(rows + C + K)
    dev = torch.device("cuda")
    mean = torch.randn(C, generator=g).double()
    inv = torch.exp(0.5 * torch.randn(C, generator=g)).double()
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)), (0.2 * torch.randn(C, generator=g))
    stats = torch.cat([mean, inv]).float()
    scale32 = gamma * stats[C:]                                  # what catseg_bn_finalize stores: fp32 gamma * invstd
    # y such that no normalised value sits within rounding distance of the ReLU threshold (the mask decision is then the same in fp32 and fp64)
    zt = torch.randn(rows, C, generator=g).double()
    zt = torch.where(zt.abs() < 1e-3, torch.full_like(zt, 1e-3) * torch.where(zt < 0, -1.0, 1.0), zt)
    y = (stats[:C].double() + (zt - beta.double()) / scale32.double()).float()
    wh = (torch.randn(K, C, generator=g) / C ** 0.5)
    bh = torch.randn(K, generator=g)
    y_d = y.view(B, H, W, C).to(dev)
    # ---- forward
    out = ops.head_fwd(y_d, stats[:C].to(dev), scale32.to(dev), beta.to(dev), wh.to(dev), bh.to(dev), K, 32)
    torch.cuda.synchronize()
    assert out.shape == (B, H, W, K) and ops.ld_of(out) == 32
    y64 = y.double()
    z64 = torch.relu((y64 - stats[:C].double()) * scale32.double() + beta.double())
    assert float(((y64 - stats[:C].double()) * scale32.double() + beta.double()).abs().min()) > 1e-4
    ref = z64 @ wh.double().t() + bh.double()
    got = out.reshape(rows, K).cpu().double()
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max() + z64.abs().max()), float((got - ref).abs().max())
    pad = out.as_strided((rows, 32), (32, 1))[:, K:]
    assert float(pad.abs().max()) == 0.0 if K < 32 else True
    # ---- backward: garbage (NaN) in the padding columns of the logits gradient must not matter
    dl = torch.full((B, H, W, 32), float("nan"))
    dl[..., :K] = torch.randn(B, H, W, K, generator=g) * 3e-6
    dl_d = dl.to(dev)[..., :K]
    dwh, dbh = torch.full((K, C), float("nan"), device=dev), torch.full((K,), float("nan"), device=dev)
    dgam, dbet, dbias = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.full((C,), float("nan"), device=dev)
    blk, sc = ops.head_backward(dl_d, y_d, stats.to(dev), gamma.to(dev), beta.to(dev), wh.to(dev), dwh, dbh, dgam, dbet, dbias)
    torch.cuda.synchronize()
    dl64 = dl[..., :K].reshape(rows, K).double()
    gz = (dl64 @ wh.double()) * (z64 > 0)
    xh = (y64 - stats[:C].double()) * stats[C:].double()
    sg, sgx = gz.sum(0), (gz * xh).sum(0)
    dy64 = (gamma * stats[C:]).double() * (gz - sg / rows - xh * (sgx / rows))
    tol = lambda r: 3e-6 * float(r.abs().max())
    assert float((dwh.cpu().double() - dl64.t() @ z64).abs().max()) <= tol(dl64.t() @ z64) + 1e-6 * float((dl64.abs().t() @ z64.abs()).max())
    assert float((dbh.cpu().double() - dl64.sum(0)).abs().max()) <= 1e-6 * float(dl64.abs().sum(0).max())
    assert float((dbet.cpu().double() - sg).abs().max()) <= 1e-6 * float(gz.abs().sum(0).max())
    assert float((dgam.cpu().double() - sgx).abs().max()) <= 1e-6 * float((gz * xh).abs().sum(0).max())
    v, e = _planes_to_f64(blk, sc, C)
    amax = float(dy64.abs().max())
    bound = np.frombuffer(np.int32(int(sc.cpu()[0])).tobytes(), dtype=np.float32)[0]
    assert bound >= amax and float(bound) * 2.0 ** e < 2.0 ** 15
    err = (v - dy64).abs()
    assert float(err.max()) <= 4e-6 * amax, (float(err.max()), amax)
    assert float((dbias.cpu().double() - dy64.sum(0)).abs().max()) <= 2e-6 * float(dy64.abs().sum(0).max())
    # deterministic: a second call reproduces every output bit for bit
    dwh2, dbh2 = torch.empty_like(dwh), torch.empty_like(dbh)
    dgam2, dbet2 = torch.empty_like(dgam), torch.empty_like(dbet)
    blk2, sc2 = ops.head_backward(dl_d, y_d, stats.to(dev), gamma.to(dev), beta.to(dev), wh.to(dev), dwh2, dbh2, dgam2, dbet2, None)
    torch.cuda.synchronize()
    assert torch.equal(blk, blk2) and torch.equal(sc, sc2) and torch.equal(dwh, dwh2) and torch.equal(dbh, dbh2)
    assert torch.equal(dgam, dgam2) and torch.equal(dbet, dbet2)


def _net(with_bias, Cin, Cout, k, K):
    from miccai2021_cataract_semantic_segmentation_amd.engine import BatchNorm2d, Conv2d, EngineNet, conv_bn_act

    class Net(EngineNet):
        def __init__(self):
            super().__init__()
            self.pre = Conv2d(Cin, Cin, 1, bias=False)
            self.pre_bn = BatchNorm2d(Cin)
            self.conv = Conv2d(Cin, Cout, k, 1, k // 2, bias=with_bias)
            self.bn = BatchNorm2d(Cout)
            self.head = Conv2d(Cout, K, 1, 1, 0, bias=True)

        def _body(self, cx, x):
            t = conv_bn_act(cx, x.permute(0, 2, 3, 1).contiguous(), self.pre, self.pre_bn)      # (NCHW API, NHWC inside)
            return [conv_bn_act(cx, t, self.conv, self.bn, head=self.head)]
    return Net


@pytest.mark.parametrize("case", [(True, 208, 256, 3, 28), (False, 256, 128, 1, 8)])      # (K % 4 == 0: the separate-pass route takes the dense output gradient as it is)
def test_layer_through_the_engine_matches_the_separate_passes(ops, case):
    """conv -> BatchNorm -> ReLU -> classifier on the head layers' route (thresholds lowered as in tests/test_heads_dy_planes_gpu.py): logits and
    every gradient of the fused route against BatchNorm apply + classifier as separate layers"""
    with_bias, Cin, Cout, k, K = case
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.HEAD_FUSE,
             ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS)
    try:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = "bf16x3", 1, 64, 32, 1, 1
        ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS = 64, 64 * 64, 64
        torch.manual_seed(5)
        net = _net(with_bias, Cin, Cout, k, K)().cuda().train()
        x = torch.randn(2, Cin, 24, 40, device="cuda")
        gout = torch.randn(2, K, 24, 40, device="cuda") * 1e-3
        res, outs = {}, {}
        for mode in (True, False):
            ops.HEAD_FUSE = mode
            ops.release_b3_cache()
            net.zero_grad()
            ops.PROFILE = []
            out = net(x)
            out = out[0] if isinstance(out, (tuple, list)) else out
            out.backward(gout)
            torch.cuda.synchronize()
            kinds = [p[0] for p in ops.PROFILE]
            ops.PROFILE = None
            assert "wgrad_h2" in kinds and "dgrad_h2" in kinds, kinds
            assert ("hbm:head_fwd" in kinds and "hbm:head_backward" in kinds) == mode, kinds
            res[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
            outs[mode] = out.detach().clone()
        a, b = outs[True].double(), outs[False].double()
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), float((a - b).abs().max())
        for n in res[True]:
            a, b = res[True][n].double(), res[False][n].double()
            scale = float(b.abs().max())
            if n == "conv.bias":       # rounding noise around 0 in both routes: compare at the scale of the weight gradient's column mass
                assert float((a - b).abs().max()) <= 1e-5 * float(res[False]["conv.weight"].abs().sum() / b.numel() + scale)
                continue
            assert float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()), scale)
    finally:
        (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.HEAD_FUSE,
         ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS) = saved
        ops.PROFILE = None
