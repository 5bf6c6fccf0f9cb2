"""OhemCrossEntropy on the HIP path (catseg_ohem_cross_entropy through the C ABI) against the reference fixtures
and against the CPU oracle at sizes where the radix select has to walk all three digits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _run(logits, target, cfg):
    from miccai2021_cataract_semantic_segmentation_amd.losses import OhemCrossEntropy
    x = logits.cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_()   # NHWC storage, NCHW view
    loss = OhemCrossEntropy(cfg)(x, target.cuda())
    loss.backward()
    return float(loss), x.grad.cpu()


def test_ohem_matches_reference_fixtures(golden):
    _need_gpu()
    g = golden("ohem")
    for name in "abcd":
        exp, mk, th = g[name + "_cfg"]
        cfg = {"experiment": int(exp)}
        if mk >= 0:
            cfg["min_kept"] = int(mk)
        if th >= 0:
            cfg["thresh"] = float(th)
        loss, grad = _run(T(g[name + "_logits"]), T(g[name + "_target"]), cfg)
        assert abs(loss - float(g[name + "_loss"])) < 2e-6 * max(1.0, abs(float(g[name + "_loss"]))), name
        # the selected set must be IDENTICAL (a pixel is selected iff its gradient row is non-zero)
        sel = grad.abs().sum(1) > 0
        ref_sel = T(g[name + "_grad"]).abs().sum(1) > 0
        assert torch.equal(sel, ref_sel), name
        np.testing.assert_allclose(grad.numpy(), g[name + "_grad"], atol=2e-7, rtol=1e-4)


@pytest.mark.parametrize("min_kept,thresh", [(100000, 0.7), (30000, 0.2), (5, 0.9), (10 ** 7, 0.0)])
def test_ohem_vs_oracle_large(min_kept, thresh):
    """half a million pixels, blob labels with ignore regions: threshold digits differ from the fixture cases"""
    _need_gpu()
    from oracle import losses as OL
    gen = torch.Generator().manual_seed(min_kept % 97)
    B, K, H, W = 2, 25, 384, 640
    lbl = torch.randint(0, 26, (B, H // 16, W // 16), generator=gen).repeat_interleave(16, 1).repeat_interleave(16, 2)
    onehot = torch.nn.functional.one_hot(lbl.clamp(max=24), K).permute(0, 3, 1, 2).float()
    logits = torch.randn(B, K, H, W, generator=gen) * 2 + onehot * torch.rand(B, 1, H, W, generator=gen) * 6
    x = logits.clone().requires_grad_()
    ref = OL.ohem_cross_entropy(x, lbl, 3, thresh=thresh, min_kept=min_kept)
    ref.backward()
    loss, grad = _run(logits, lbl, {"experiment": 3, "min_kept": min_kept, "thresh": thresh})
    assert abs(loss - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    sel, ref_sel = grad.abs().sum(1) > 0, x.grad.abs().sum(1) > 0
    # fp32 exp/div differ by an ulp between the two implementations: pixels exactly AT the threshold may flip
    assert (sel != ref_sel).sum().item() <= 8
    same = (sel == ref_sel).unsqueeze(1).expand_as(grad)
    np.testing.assert_allclose(grad[same].numpy(), x.grad[same].numpy(), atol=1e-9, rtol=2e-3)


def test_ohem_all_ignored_is_nan_and_two_scale():
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    logits = torch.randn(1, 25, 8, 8)
    loss, grad = _run(logits, torch.full((1, 8, 8), 25), {"experiment": 3})
    assert np.isnan(loss) and float(grad.abs().max()) == 0.0
    from oracle import losses as OL
    gen = torch.Generator().manual_seed(2)
    a, b = torch.randn(2, 17, 32, 48, generator=gen) * 2, torch.randn(2, 17, 32, 48, generator=gen) * 2
    lbl = torch.randint(0, 18, (2, 32, 48), generator=gen)
    crit = TwoScaleLoss({"experiment": 2, "interm": {"name": "OhemCrossEntropy", "args": [], "min_kept": 500},
                         "final": {"name": "OhemCrossEntropy", "args": [], "min_kept": 500}})
    got = crit(a.cuda(), b.cuda(), lbl.cuda())
    want = OL.ohem_cross_entropy(b, lbl, 2, min_kept=500) + 0.4 * OL.ohem_cross_entropy(a, lbl, 2, min_kept=500)
    assert abs(float(got) - float(want)) < 1e-5
