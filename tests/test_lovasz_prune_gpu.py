"""Active-set pruning in catseg_lovasz_softmax (only elements that can precede the last foreground pixel are sorted)
must be BIT-identical to sorting everything, for untrained (uniform), confident and adversarial (ties, tiny classes) logits."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(logits, labels):
    from miccai2021_cataract_semantic_segmentation_amd import ops, _lib
    out = []
    for prune in (0, 1):
        _lib.lib.catseg_debug_set_lovasz_prune(prune)
        try:
            dl = torch.empty_like(logits)
            loss = ops.lovasz_softmax(logits, labels, 1.0, dl)
            out.append((loss.clone(), dl.clone()))
        finally:
            _lib.lib.catseg_debug_set_lovasz_prune(1)
    return out


@pytest.mark.parametrize("kind", ["uniform", "confident", "ties", "one_pixel_class", "large"])
def test_pruned_equals_full_sort(kind):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import losses as OL
    g = torch.Generator().manual_seed(len(kind))
    K = 25
    P = 300000 if kind == "large" else 20000
    labels = torch.randint(0, K + 1, (P // 50,), generator=g).repeat_interleave(50)     # blobs, label K = ignore
    logits = torch.randn(P, K, generator=g) * 0.05
    if kind in ("confident", "large"):
        onehot = torch.nn.functional.one_hot(labels.clamp(max=K - 1), K).float()
        logits = logits * 20 + onehot * 9 * (torch.rand(P, 1, generator=g) > 0.1)       # 10 % hard pixels
    if kind == "ties":
        logits = (logits * 40).round() / 4                                               # many exactly equal errors
    if kind == "one_pixel_class":
        labels[labels == 7] = 8
        labels[1234] = 7
    (l0, d0), (l1, d1) = _both(logits.cuda(), labels.cuda())
    assert torch.equal(l0, l1), (float(l0), float(l1))
    assert torch.equal(d0, d1)
    # and both agree with the oracle (numpy restatement of losses/LovaszSoftmax.py)
    ref = OL.lovasz_softmax_np(logits.view(1, P, 1, K).permute(0, 3, 1, 2).numpy(), labels.view(1, P, 1).numpy())
    assert abs(float(l1) - ref) < 3e-6
