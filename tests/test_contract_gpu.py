"""Contract gaps closed in round 4, each on the HIP path through the C ABI:
  * the resize-on-mismatch branch of TwoScaleLoss / OhemCrossEntropy (losses/TwoScaleLoss.py:45-48, losses/OhemCrossEntropy.py:23-26)
    against fixtures from the REAL reference (tests/golden/make_golden_resize.py);
  * stock torch.optim.Adam(model.parameters()) in the reference manager's step order (managers/OCRNet_Manager.py:80-90) against FusedAdam;
  * an amax record that no longer bounds its tensor (something was accumulated into it) is dropped, never consumed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _dev(x):
    """NHWC storage, NCHW view (what the engine's networks return), requires grad"""
    return x.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_()


@pytest.mark.parametrize("name,lname,exp,extra", [("ts_lovasz", "LovaszSoftmax", 3, {}), ("ts_ce", "CrossEntropyLoss", 2, {}),
                                                  ("ts_ohem", "OhemCrossEntropy", 3, {"min_kept": 150, "thresh": 0.6})])
def test_two_scale_loss_upsamples_low_resolution_intermediate_logits(golden, name, lname, exp, extra):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    g = golden("losses_resize")
    cfg = {"experiment": exp, "interm": dict({"name": lname, "args": [], "weight": 0.4}, **extra),
           "final": dict({"name": lname, "args": [], "weight": 1.0}, **extra)}
    i, f = _dev(T(g[name + "_interm"])), _dev(T(g[name + "_final"]))
    loss = TwoScaleLoss(cfg)(i, f, T(g[name + "_target"]).cuda())
    loss.backward()
    want = float(g[name + "_loss"])
    assert abs(float(loss) - want) < 3e-6 * max(1.0, abs(want)), (float(loss), want)
    np.testing.assert_allclose(i.grad.cpu().numpy(), g[name + "_ginterm"], atol=3e-7, rtol=2e-4)
    np.testing.assert_allclose(f.grad.cpu().numpy(), g[name + "_gfinal"], atol=3e-7, rtol=2e-4)


def test_ohem_upsamples_low_resolution_scores(golden):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.losses import OhemCrossEntropy
    g = golden("losses_resize")
    s = _dev(T(g["ohem_score"]))
    loss = OhemCrossEntropy({"experiment": 2, "min_kept": 500, "thresh": 0.5})(s, T(g["ohem_target"]).cuda())
    loss.backward()
    assert abs(float(loss) - float(g["ohem_loss"])) < 3e-6 * max(1.0, abs(float(g["ohem_loss"])))
    sel, ref_sel = s.grad.cpu().abs().sum(1) > 0, T(g["ohem_grad"]).abs().sum(1) > 0
    assert (sel != ref_sel).sum().item() == 0         # the same low-resolution pixels receive gradient
    np.testing.assert_allclose(s.grad.cpu().numpy(), g["ohem_grad"], atol=3e-7, rtol=2e-4)


def test_stock_torch_adam_in_the_reference_loop_matches_fused_adam():
    """`torch.optim.Adam(model.parameters(), lr)` (managers/BaseManager.py:441) keeps working on the engine's parameters:
    three steps in the reference's order (zero_grad -- set_to_none, the torch default --, forward, loss, backward, step).  Every step is
    compared with ONE FusedAdam step from the identical state (weights, BatchNorm buffers and Adam moments copied over through the two
    state_dict formats): this random-weight 64 x 96 network amplifies a 1e-7 parameter difference into a 1 % gradient difference within one
    step (tools/adam_dbg.py), so free-running trajectories of ANY two Adam implementations part after two steps."""
    _need_gpu()
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    cfgm = {"backbone": "resnet50", "out_stride": 8, "pretrained": False}
    g = torch.Generator().manual_seed(3)
    xs = [torch.rand(2, 3, 64, 96, generator=g).cuda() for _ in range(3)]
    ls = [torch.randint(0, 26, (2, 8, 12), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2).cuda() for _ in range(3)]

    def make():
        model = OCRNet(dict(cfgm), 3)
        model.load_state_dict(fill_state(spec_of(model.state_dict()), 5))
        model.cuda().train()
        return model
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    ma, mb = make(), make()
    stock, fused = torch.optim.Adam(ma.parameters(), lr=1e-3), FusedAdam(mb, lr=1e-3)
    start = {k: v.detach().clone() for k, v in ma.state_dict().items()}
    for i, (x, l) in enumerate(zip(xs, ls)):
        mb.load_state_dict(ma.state_dict())                  # identical weights, BatchNorm buffers ...
        if i:
            fused.load_state_dict(stock.state_dict())        # ... and Adam moments / step count (torch.optim.Adam's own format)
        losses = []
        for model, opt in ((ma, stock), (mb, fused)):
            opt.zero_grad()
            interm, final = model(x)
            loss = crit(interm, final, l)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert losses[0] == losses[1], (i, losses)
        sa, sb = ma.state_dict(), mb.state_dict()
        for k in sa:
            if sa[k].dtype.is_floating_point:
                d = float((sa[k] - sb[k]).abs().max())
                assert d <= 1e-6, (i, k, d)                  # the update itself is ~1e-3 per step
            else:
                assert torch.equal(sa[k], sb[k]), (i, k)
    moved = max(float((v - start[k]).abs().max()) for k, v in ma.state_dict().items() if v.dtype.is_floating_point and "running" not in k)
    assert 2e-3 < moved < 4e-3, moved       # three Adam steps of lr 1e-3 under the stock optimiser


def test_stale_amax_record_is_dropped_not_consumed():
    """a record says max|x| = 1; then 1e6 is accumulated into x in place.  Consumed as it stands, the record would scale x to 2^14 x 1e6 and
    overflow the fp16 planes (inf / nan).  In-place consumers clear the record (ops.drop_amax): the trunk convolution then takes the
    three-plane bf16 kernel and its result is finite and right."""
    _need_gpu()
    import torch.nn.functional as F
    from miccai2021_cataract_semantic_segmentation_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    x = torch.rand(1, 24, 40, 48, generator=g).to(dev)
    x._amax = ops.new_amax(dev)
    x._amax[0:1] = x.abs().max().reshape(1).view(torch.int32)
    big = (torch.rand(1, 24, 40, 48, generator=g) * 1e6).to(dev)
    w = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    saved = (ops.TRUNK, ops.PRECISION, ops.DCONV3_MIN_ROWS)
    ops.TRUNK, ops.PRECISION, ops.DCONV3_MIN_ROWS = "f16x2", "bf16x3", 1
    try:
        ops.PROFILE = []
        y0 = ops.conv_fwd(x, w, None, 48, 3, 3, 1, 1, 1)                 # the record is valid: two fp16 planes
        ops.axpy(big, x, 1.0, True)                                      # x += big, in place
        assert ops.amax_of(x) is None
        y1 = ops.conv_fwd(x, w, None, 48, 3, 3, 1, 1, 1)
        kinds = [k for k, *_ in ops.PROFILE]
        assert kinds == ["fwd_d3h", "fwd_d3"], kinds
        # the engine's accumulation points: a gradient buffer that arrives with a record and is accumulated into
        from miccai2021_cataract_semantic_segmentation_amd.engine import Ctx
        cx = Ctx(True, True)
        t = torch.zeros(1, 4, 4, 48, device=dev)
        gbuf = torch.ones(1, 4, 4, 48, device=dev)
        gbuf._amax = ops.new_amax(dev)
        cx.give(t, gbuf)
        buf, acc = cx.dest(t)
        assert acc and buf is gbuf and ops.amax_of(buf) is None
    finally:
        ops.PROFILE = None
        ops.TRUNK, ops.PRECISION, ops.DCONV3_MIN_ROWS = saved
        ops.release_b3_cache()
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    assert torch.isfinite(y1).all()
    assert float((y1.cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.isfinite(y0).all()
