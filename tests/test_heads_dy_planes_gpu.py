"""The head layers' BatchNorm backward writing dy straight as the blocked fp16 x 2 planes of its two consumers (csrc/norm.hip:
bn_bwd_apply_h2_kernel, ops.bn_backward_h2 / h2_dy_route; models/OCR.py:72-89, 326-333 of the reference are the layers):
the kernel against the fp32 route + catseg_split2h, the layer through the engine against the previous route, and the route predicate
against what conv_bwd_weight / conv_bwd_data select."""
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


@pytest.mark.parametrize("shape", [(2, 20, 24, 128), (1, 37, 19, 64), (3, 16, 16, 512)])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_backward_h2_vs_fp32_route(ops, shape, relu):
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B * 131 + C)
    dev = torch.device("cuda")
    y = (torch.randn(B, H, W, C, generator=g) * torch.exp(torch.randn(C, generator=g)) + torch.randn(C, generator=g)).to(dev)
    dz = (torch.randn(B, H, W, C, generator=g) * 3e-6).to(dev)
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)).to(dev), (0.2 * torch.randn(C, generator=g)).to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats, _ = ops.bn_train_stats(y, gamma, 1e-5, 0.1, rm, rv)
    # the fp32 route: dy, dgamma, dbeta; the bias gradient = column sums of dy
    dg0, db0 = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dy = ops.bn_backward(dz, None, y, stats, gamma, relu, dg0, db0, beta=beta)
    # the planes route
    dg1, db1, dbias = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.full((C,), float("nan"), device=dev)
    blk, scale = ops.bn_backward_h2(dz, y, stats, gamma, relu, dg1, db1, beta, dbias)
    torch.cuda.synchronize()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1), "dgamma / dbeta differ from the fp32 route"
    v, e = _planes_to_f64(blk, scale, C)
    ref = dy.cpu().double().reshape(-1, C)
    amax = float(ref.abs().max())
    bound = np.frombuffer(np.int32(int(scale.cpu()[0])).tobytes(), dtype=np.float32)[0]
    assert bound >= amax, "the bound of dy is below its true maximum"
    # (max over the channels of |gamma invstd| times the GLOBAL max|g|: on this data -- per-channel spreads of e^+-3 -- up to ~10x the true maximum;
    #  2^k of looseness costs k bits of the l plane's subnormal threshold, 2^(k - 39) of the largest element: the accuracy check below sees it)
    assert bound <= 64.0 * amax, "the bound of dy is looser than 64x (%g vs %g)" % (bound, amax)
    assert float(bound) * 2.0 ** e < 2.0 ** 15
    err = (v - ref).abs()
    tol = torch.maximum(ref.abs() * 2.0 ** -22, torch.full_like(ref, 2.0 ** -25 * 2.0 ** -e))
    assert bool((err <= tol).all()), float((err / tol).max())
    # bias gradient: column sums of dy (mathematically 0: compare against the fp64 sum at the scale of sum |dy|)
    s64 = ref.sum(0)
    assert float((dbias.cpu().double() - s64).abs().max()) <= 1e-6 * float(ref.abs().sum(0).max())
    # without a bias the entry writes nothing there
    blk2, scale2 = ops.bn_backward_h2(dz, y, stats, gamma, relu, dg1, db1, beta, None)
    assert torch.equal(blk2, blk) and torch.equal(scale2, scale)


def _layer(ops, with_bias, Cin, Cout, k):
    from miccai2021_cataract_semantic_segmentation_amd.engine import BatchNorm2d, Conv2d, EngineNet, conv_bn_act

    class Net(EngineNet):
        def __init__(self):
            super().__init__()
            self.pre = Conv2d(Cin, Cin, 1, bias=False)
            self.pre_bn = BatchNorm2d(Cin)
            self.conv = Conv2d(Cin, Cout, k, 1, k // 2, bias=with_bias)
            self.bn = BatchNorm2d(Cout)

        def _body(self, cx, x):
            t = conv_bn_act(cx, x.permute(0, 2, 3, 1).contiguous(), self.pre, self.pre_bn)      # (NCHW API, NHWC inside)
            return [conv_bn_act(cx, t, self.conv, self.bn)]
    return Net


@pytest.mark.parametrize("case", [(True, 208, 256, 3), (False, 256, 128, 1)])
def test_layer_through_the_engine_matches_the_previous_route(ops, case):
    """conv -> BatchNorm -> ReLU on the f16x2 kernels (thresholds lowered as tests/conftest.py does): gradients with dy as planes against dy as
    fp32 + split pass; both runs take the f16x2 backward kernels"""
    with_bias, Cin, Cout, k = case
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.HEAD_DY_PLANES,
             ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS)
    try:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = "bf16x3", 1, 64, 32, 1, 1
        ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS = 64, 64 * 64, 64
        torch.manual_seed(3)
        net = _layer(ops, with_bias, Cin, Cout, k)().cuda().train()
        x = torch.randn(2, Cin, 24, 40, device="cuda")
        gout = torch.randn(2, Cout, 24, 40, device="cuda") * 1e-3
        res = {}
        for mode in (True, False):
            ops.HEAD_DY_PLANES = mode
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
            res[mode] = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
        for n in res[True]:
            a, b = res[True][n].double(), res[False][n].double()
            scale = float(b.abs().max())
            if n == "conv.bias":       # rounding noise around 0 in both routes: compare at the scale of the weight gradient's column mass
                assert float((a - b).abs().max()) <= 1e-5 * float(res[False]["conv.weight"].abs().sum() / b.numel() + scale)
                continue
            assert float((a - b).abs().max()) <= 2e-5 * scale, (n, float((a - b).abs().max()), scale)
    finally:
        (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.HEAD_DY_PLANES,
         ops.B3_1X1_MIN_DIM, ops.B3_1X1_MIN_PROD, ops.B3_1X1_MIN_ROWS) = saved
        ops.PROFILE = None


def test_route_predicate_on_the_bench_shapes(ops):
    """the three head layers of the bench model (8 x 136 x 240: 720 -> 512 3 x 3 twice, 1024 -> 512 1 x 1) take the planes route under the
    production thresholds; trunk-shaped and small layers do not (the predicate reads shapes, never data)"""
    dev = torch.device("cuda")
    for Cin, Cout, k, want in [(720, 512, 3, True), (1024, 512, 1, True), (48, 48, 3, False), (512, 256, 1, False), (256, 512, 1, False)]:
        x, y = torch.empty(8, 136, 240, Cin, device=dev), torch.empty(8, 136, 240, Cout, device=dev)
        w = torch.empty(Cout, Cin, k, k, device=dev)
        assert ops.h2_dy_route(x, y, w, k, k, 1, k // 2, 1, 1, True) == want, (Cin, Cout, k)


def test_concat_bilinear_planes_vs_fp32_concat_and_split(ops):
    """ops.concat_bilinear_h2 (one launch: interpolate, split, write blocked planes) against bilinear_fwd into channel slices + catseg_split2h"""
    g = torch.Generator().manual_seed(9)
    dev = torch.device("cuda")
    B, H, W = 2, 24, 40
    ys = []
    for C, sdiv, mag in ((48, 1, 1.0), (96, 2, 3.0), (192, 4, 0.2), (384, 8, 40.0)):
        y = (torch.randn(B, H // sdiv, W // sdiv, C, generator=g) * mag).to(dev)
        rec = ops.new_amax(dev)
        rec[0] = int(np.frombuffer(np.float32(float(y.abs().max())).tobytes(), dtype=np.int32)[0])
        y._amax = rec
        ys.append(y)
    cat = torch.empty(B, H, W, 720, device=dev)
    c0 = 0
    for y in ys:
        C = y.shape[-1]
        if y.shape[1] == H:
            cat[..., c0:c0 + C].copy_(y)
        else:
            ops.bilinear_fwd(y, H, W, False, out=cat[..., c0:c0 + C])
        c0 += C
    blk, sc = ops.concat_bilinear_h2(ys, H, W)
    torch.cuda.synchronize()
    amax = max(float(y.abs().max()) for y in ys)
    assert np.frombuffer(np.int32(int(sc.cpu()[0])).tobytes(), dtype=np.float32)[0] == np.float32(amax)
    v, e = _planes_to_f64(blk, sc, 720)
    ref = cat.cpu().double().reshape(-1, 720)
    err = (v - ref).abs()
    # 22-bit planes of the interpolated value; the interpolation may contract its multiply-adds differently in the two kernels: an ulp of its
    # largest TERM (a result that cancels to ~0 between neighbours carries that absolute error)
    tol = torch.maximum(ref.abs() * 2.0 ** -22, torch.full_like(ref, 2.0 ** -25 * 2.0 ** -e))
    c0 = 0
    for y in ys:
        C = y.shape[-1]
        tol[:, c0:c0 + C] += 2.0 ** -23 * float(y.abs().max())
        c0 += C
    assert bool((err <= tol).all()), float((err / tol).max())
    # branch 0 is a copy: its planes are exactly the split of its values
    b0, s0 = ops.split2h_blocked(cat[..., :48].contiguous())
    if int(s0.cpu()[1]) == e:
        assert torch.equal(blk[:, :3], b0)


def test_ocrnet_hrnet_step_with_and_without_the_concat_planes(ops):
    """a small OCRNet-HRNet training step (thresholds lowered so that the head layers take the f16x2 kernels) with the trunk output as planes
    only (A) against the fp32 concatenation + split pass (B): logits to 2e-5; gradients against the step's own numerical sensitivity --
    B re-run on an image perturbed by 1e-7 relative (C): at 2 x 24 x 40 head pixels the BatchNorm / attention backward passes amplify such
    a perturbation to 1e-2 of some head gradients, so |A - B| is held to a small multiple of |B - C|, tensor by tensor"""
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    saved = (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.CONCAT_PLANES)
    try:
        ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS = "bf16x3", 1, 64, 32, 1, 1
        torch.manual_seed(5)
        net = OCRNet({"backbone": "hrnet48", "pretrained": False}, 3).cuda().train()
        x = torch.randn(2, 3, 96, 160, device="cuda")
        x2 = x * (1 + 1e-7 * torch.randn_like(x))
        r1, r2 = torch.randn(2, 25, 96, 160, device="cuda"), torch.randn(2, 25, 96, 160, device="cuda")
        res = {}
        for tag, mode, inp in (("A", True, x), ("B", False, x), ("C", False, x2)):
            ops.CONCAT_PLANES = mode
            net.zero_grad()
            ops.PROFILE = []
            interm, final = net(inp)
            (final * r1).mean().add(0.4 * (interm * r2).mean()).backward()
            torch.cuda.synchronize()
            kinds = [p[0] for p in ops.PROFILE]
            ops.PROFILE = None
            assert kinds.count("fwd_h2") >= 2 and kinds.count("wgrad_h2") >= 2, kinds
            assert ("split3" in kinds), kinds
            res[tag] = (final.detach().clone(), interm.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
        for k in (0, 1):
            assert float((res["A"][k] - res["B"][k]).abs().max()) <= 2e-5 * float(res["B"][k].abs().max())
        # (the biases of the two head convolutions sit in front of a BatchNorm: their gradient is rounding noise around 0 in any implementation)
        noise = ("interm_prediction_head.0.bias", "conv_high_map.0.bias")
        bad = []
        for n in res["A"][2]:
            if n in noise:
                continue
            a, b, c = (res[t][2][n].double() for t in "ABC")
            scale = float(b.abs().max()) + 1e-30
            dab, dbc = float((a - b).abs().max()) / scale, float((b - c).abs().max()) / scale
            if dab > 4.0 * dbc + 1e-5:
                bad.append((n, dab, dbc))
        assert not bad, bad[:8]
    finally:
        (ops.PRECISION, ops.B3_MIN_TAPS, ops.B3_MIN_K, ops.B3_MIN_N, ops.B3_MIN_TILES, ops.B3_MIN_WGRAD_ROWS, ops.CONCAT_PLANES) = saved
        ops.PROFILE = None
