"""Size-independent properties at the BENCH sizes (bs 8 @ 544x960 feature maps), where the CPU oracle is too slow:
the adjoint identity ties the three convolution kernels together without any reference,

    <dy, conv(x, w)>  ==  <conv_bwd_data(dy, w), x>  ==  <conv_bwd_weight(x, dy), w>,

linearity of the forward in x, and determinism (two runs bit-identical: the split reductions have a fixed order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # name, B, H, W, Cin, Cout, k, stride, pad, dil         (paths exercised)
    ("hrnet branch 48", 8, 136, 240, 48, 48, 3, 1, 1, 1),      # 16x16x4 48-wide tiles + direct backward-weight
    ("hrnet branch 96", 8, 68, 120, 96, 96, 3, 1, 1, 1),       # 96-wide tiles + direct backward-weight (one block per filter row)
    ("hrnet fuse s2", 8, 136, 240, 48, 96, 3, 2, 1, 1),        # strided: parity-decomposed backward-data
    ("resnet l4 d4", 8, 68, 120, 512, 512, 3, 1, 4, 4),        # 128x128 / 256x128 tiles, dilation
    ("resnet 1x1", 8, 68, 120, 1024, 256, 1, 1, 0, 1),
    ("ocr head 720", 4, 136, 240, 720, 512, 3, 1, 1, 1),       # the largest layer of the HRNet step (half batch: memory of the test)
]


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_adjoint_linearity_determinism(case):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from miccai2021_cataract_semantic_segmentation_amd import ops
    _, B, H, W, Ci, Co, k, s, p, d = case
    g = torch.Generator(device="cuda").manual_seed(B * H + Ci)
    dev = torch.device("cuda")
    x = torch.randn(B, H, W, Ci, device=dev, generator=g)
    w = (torch.randn(Co, Ci, k, k, device=dev, generator=g) * (1.0 / (Ci * k * k) ** 0.5)).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, Co, k, k, s, p, d)
    dy = torch.randn(y.shape, device=dev, generator=g)
    dx = ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, d)
    dw = torch.empty_like(w)
    ops.conv_bwd_weight(x, dy, dw, None, k, k, s, p, d)
    a, b, c = dot(dy, y), dot(dx, x), dot(dw, w)
    scale = (dot(dy, dy) * dot(y, y)) ** 0.5
    assert abs(a - b) <= 2e-5 * scale and abs(a - c) <= 2e-5 * scale, (a, b, c, scale)
    # linearity in x
    x2 = torch.randn(B, H, W, Ci, device=dev, generator=g)
    y2 = ops.conv_fwd(x2, w, None, Co, k, k, s, p, d)
    y12 = ops.conv_fwd(x + x2, w, None, Co, k, k, s, p, d)
    err = float((y12 - (y + y2)).abs().max())
    assert err <= 2e-5 * float(y12.abs().max()) + 1e-5, err
    # determinism of the split reductions
    dw2 = torch.empty_like(w)
    ops.conv_bwd_weight(x, dy, dw2, None, k, k, s, p, d)
    assert torch.equal(dw, dw2)
    assert torch.equal(ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, d), dx)


def test_train_step_fullsize_is_deterministic_and_finite():
    """one full-size OCRNet-HRNet-W48 training step (the bench workload) twice from the same state: identical loss and
    gradient buffer bit for bit (no atomics anywhere on the path), everything finite"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import bench
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    torch.manual_seed(0)
    dev = torch.device("cuda")
    model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3).to(dev).train()
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    img, lbl = bench.synth_batch(4, 544, 960, 25, 1, dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    outs = []
    for _ in range(2):
        model.load_state_dict(state)
        model.flat().grad.zero_()
        i, f = model(img)
        loss = crit(i, f, lbl)
        loss.backward()
        outs.append((loss.detach().clone(), model.flat().grad.clone()))
    assert torch.isfinite(outs[0][0]) and bool(torch.isfinite(outs[0][1]).all())
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
