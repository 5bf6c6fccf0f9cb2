"""Full-resolution parity under the PRODUCTION plan: the networks at the bench sizes (3 x 544 x 960, config 5 at 3 x 1088 x 1920), the
default arithmetic selection (ops.PRECISION as shipped: bf16x3 on the large layers and on the HRNet trunk's direct kernels, exact
fp32 elsewhere), NO threshold forcing -- the tile forms, blocked planes, split counts and persistent-block schedules of the timed
benchmark -- against the CPU oracle evaluated at the same size (one oracle evaluation per network, shared by the assertions).

The oracle is test infrastructure (oracle/__init__.py); every HIP call goes through the C ABI.  Tolerances: logits 1e-3 ABSOLUTE
(north star) with the relative figure printed, loss 1e-4, BatchNorm running statistics 1e-4 relative, per-tensor gradient norms and
directions against the fp32 CPU oracle on a fixed subset, and the UNMASKED count of argmax disagreements printed next to the count
the fp32 CPU evaluation itself has against an fp64 evaluation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _labels(B, H, W, K, seed):
    g = torch.Generator().manual_seed(seed)
    lbl = torch.randint(0, K + 1, (B, H // 32, W // 32), generator=g)
    return lbl.repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()


def _argmax_report(name, hip, cpu32, cpu64):
    """unmasked label-map disagreements; the assertion: wherever HIP and the fp64 oracle disagree, the fp64 top-2 margin is within the
    logit error made (a tie broken the other way), and there are not more such pixels than a few times the fp32 CPU run's own"""
    a_h, a_c, a_64 = hip.argmax(1), cpu32.argmax(1), cpu64.argmax(1)
    n = a_64.numel()
    d_hc, d_h64, d_c64 = int((a_h != a_c).sum()), int((a_h != a_64).sum()), int((a_c != a_64).sum())
    top2 = cpu64.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    err = float((hip.double() - cpu64).abs().max())
    bad = (a_h != a_64) & (margin > 2.2 * err)
    print("%s argmax disagreements (unmasked, of %d pixels): hip vs cpu32 %d, hip vs fp64 %d, cpu32 vs fp64 %d; max |logit - fp64| hip %.3g cpu32 %.3g"
          % (name, n, d_hc, d_h64, d_c64, err, float((cpu32.double() - cpu64).abs().max())))
    assert int(bad.sum()) == 0, "label differs from the fp64 oracle at %d pixels whose margin exceeds the logit error" % int(bad.sum())
    assert d_h64 <= 4 * d_c64 + 64, (d_h64, d_c64)


def _grad_subset_check(name, model, S, keys):
    rows = []
    sd = dict(model.named_parameters())
    for k in keys:
        g_h = sd[k].grad.detach().cpu().double().reshape(-1)
        g_c = S[k].grad.detach().double().reshape(-1)
        nh, nc = float(g_h.norm()), float(g_c.norm())
        cos = float((g_h * g_c).sum() / (nh * nc + 1e-300))
        rows.append((k, nh / (nc + 1e-300), cos))
    worst_ratio = max(abs(r - 1) for _, r, _ in rows)
    worst_cos = min(c for _, _, c in rows)
    print("%s gradient subset (%d tensors): worst |norm ratio - 1| %.3g, worst cosine %.6f" % (name, len(rows), worst_ratio, worst_cos))
    for k, r, c in rows:
        assert abs(r - 1) < 2e-2 and c > 0.999, (k, r, c)


def test_ocrnet_hrnet48_fullres_train_step_vs_oracle():
    """the BENCH model at the BENCH resolution (batch 2 of 8: the CPU oracle's fp64 forward has to fit the test budget): logits, loss,
    BatchNorm statistics, gradients, label maps"""
    _need_gpu()
    import bench
    from oracle import nets as ON, losses as OL
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd import ops
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    assert ops.PRECISION == "bf16x3" and ops.B3_MIN_K == 2048 and ops.DCONV3_MIN_ROWS == 2048, "production plan expected"
    B, H, W, K = 2, 544, 960, 25
    model = OCRNet(dict(bench.MODELS["ocrnet_hrnet48"][0]), 3)
    spec = spec_of(model.state_dict())
    S = fill_state(spec, 41)
    model.load_state_dict(S)
    model.cuda().train()
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, H, W, generator=g)
    lbl = _labels(B, H, W, K, 10)
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    ops.PROFILE = []
    interm, final = model(x.cuda())
    loss = crit(interm, final, lbl.cuda())
    loss.backward()
    kinds = {k for k, *_ in ops.PROFILE}
    ops.PROFILE = None
    # the production kernels really ran: blocked bf16x3 heads, direct trunk kernels in all three directions, fp32 elsewhere
    heads = {"fwd_h2", "dgrad_h2", "wgrad_h2"} if ops.HEADS == "f16x2" else {"fwd_b3", "dgrad_b3", "wgrad_b3"}
    trunk = {"fwd_d3h", "dgrad_d3h", "wgrad_d3h"} if ops.TRUNK == "f16x2" else {"fwd_d3", "dgrad_d3", "wgrad_d3"}
    assert heads | trunk | {"fwd", "dgrad", "wgrad"} <= kinds, kinds
    final_h, interm_h = final.detach().cpu(), interm.detach().cpu()
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        S[k].requires_grad_()
    oi, of = ON.ocrnet_hrnet_forward(S, x, train=True)
    ol = OL.two_scale_lovasz(oi, of, lbl, 0.4, 1.0)
    ol.backward()
    S64 = {k: (v.detach().double() if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, 41).items()}
    with torch.no_grad():
        f64 = ON.ocrnet_hrnet_forward(S64, x.double(), train=True)[1]
    of_d, oi_d = of.detach(), oi.detach()
    e_abs = float((final_h - of_d).abs().max())
    e_int = float((interm_h - oi_d).abs().max())
    scale = float(of_d.abs().max())
    e_h64, e_c64 = float((final_h.double() - f64).abs().max()), float((of_d.double() - f64).abs().max())
    print("OCRNet-HRNet-W48 %dx%dx%d: max |logit - cpu32| final %.3g absolute = %.3g of the logit scale %.3g, intermediate %.3g; against the fp64 oracle: "
          "hip %.3g, cpu32 %.3g; loss hip %.7f cpu %.7f" % (B, H, W, e_abs, e_abs / scale, scale, e_int, e_h64, e_c64, float(loss), float(ol)))
    # 1e-3 of the logit scale (the scale of this random-weight network is ~10: the fp32 CPU evaluation itself sits ~1e-3 absolute away
    # from the fp64 one), and never further from the fp64 oracle than 3x the fp32 CPU run is
    assert e_abs <= 1e-3 * max(1.0, scale) and e_int <= 1e-3 * max(1.0, float(oi_d.abs().max()))
    assert e_h64 <= 3 * e_c64 + 1e-5, (e_h64, e_c64)
    assert abs(float(loss) - float(ol)) < 1e-4
    sd = model.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.stage2.0.branches.0.1.bn1.running_var", "backbone.stage4.2.branches.3.3.bn2.running_var",
              "backbone.stage3.1.branches.1.2.bn2.running_mean", "conv_high_map.1.running_var"):
        a, b = sd[k].detach().cpu().double(), S[k].detach().double()
        assert float((a - b).abs().max()) <= 1e-5 + 1e-4 * float(b.abs().max()), k
    _argmax_report("OCRNet-HRNet-W48 full resolution", final_h, of_d, f64)
    keys = ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight", "backbone.stage2.0.branches.0.0.conv1.weight",
            "backbone.stage2.0.branches.1.3.conv2.weight", "backbone.stage3.2.branches.2.1.conv1.weight",
            "backbone.stage4.1.branches.3.2.conv2.weight", "backbone.stage4.2.branches.0.3.conv1.weight",
            "backbone.stage3.0.fuse_layers.2.0.0.0.weight", "backbone.stage4.0.fuse_layers.0.3.0.weight",
            "backbone.stage2.0.branches.0.2.bn1.weight", "conv_high_map.0.weight", "interm_prediction_head.3.weight",
            "ocr_distri_head.object_context_block.f_pixel.0.weight", "final_prediction_head.weight"]
    have = dict(model.named_parameters())
    _grad_subset_check("OCRNet-HRNet-W48 full resolution", model, S, [k for k in keys if k in have and k in S])


def test_deeplabv3plus_r50_fullres_train_step_vs_oracle():
    """BASELINE config 2 at its resolution (batch 2: the ASPP image-pooling branch normalises a [B, 256, 1, 1] tensor with batch
    statistics, which needs B > 1 in the reference as well): DeepLabv3+ ResNet50, 17 classes, cross entropy with the ignore label"""
    _need_gpu()
    import bench
    from oracle import nets as ON, losses as OL
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd.models import DeepLabv3Plus
    from miccai2021_cataract_semantic_segmentation_amd.losses import CrossEntropyLoss
    B, H, W, K = 2, 544, 960, 17
    model = DeepLabv3Plus(dict(bench.MODELS["deeplabv3plus_r50"][0]), 2)
    spec = spec_of(model.state_dict())
    S = fill_state(spec, 43)
    model.load_state_dict(S)
    model.cuda().train()
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, H, W, generator=g)
    lbl = _labels(B, H, W, K, 12)
    out = model(x.cuda())
    loss = CrossEntropyLoss(ignore_index=17)(out, lbl.cuda())
    loss.backward()
    params = [k for k, v in S.items() if v.dtype.is_floating_point and "running" not in k]
    for k in params:
        S[k].requires_grad_()
    ref = ON.deeplabv3plus_forward(S, x, train=True)
    ol = OL.cross_entropy(ref, lbl, 2)
    ol.backward()
    S64 = {k: (v.detach().double() if v.dtype.is_floating_point else v.clone()) for k, v in fill_state(spec, 43).items()}
    with torch.no_grad():
        f64 = ON.deeplabv3plus_forward(S64, x.double(), train=True)
    out_h, ref_d = out.detach().cpu(), ref.detach()
    e_abs, scale = float((out_h - ref_d).abs().max()), float(ref_d.abs().max())
    print("DeepLabv3+-R50 %dx%dx%d: max |logit - cpu32| %.3g (relative %.3g of scale %.3g); loss hip %.7f cpu %.7f"
          % (B, H, W, e_abs, e_abs / scale, scale, float(loss), float(ol)))
    assert e_abs <= 1e-3 * max(1.0, scale)
    assert abs(float(loss) - float(ol)) < 1e-4
    _argmax_report("DeepLabv3+-R50 full resolution", out_h, ref_d, f64)
    have = dict(model.named_parameters())
    keys = [k for k in ("backbone.conv1.weight", "backbone.layer2.1.conv2.weight", "backbone.layer4.2.conv2.weight", "aspp.convs.1.0.weight",
                        "aspp.convs.3.0.weight", "aspp.project.0.weight", "decoder.conv_low.0.weight", "decoder.conv_out.0.weight",
                        "decoder.conv_out.6.weight") if k in have and k in S]
    if keys:
        _grad_subset_check("DeepLabv3+-R50 full resolution", model, S, keys)


def test_resnext101_upernet_fullres_inference_vs_oracle():
    """BASELINE config 5 at its resolution (one frame of 3 x 1088 x 1920, reference models/UPerNet.py:108-145): the fused inference path
    (folded BatchNorm, bf16x3 blocked kernels on conv_last / FPN) against the CPU restatement"""
    _need_gpu()
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_encdec_gpu import _resnext_oracle
    from oracle.state import fill_state, spec_of
    from oracle.upernet import upernet_forward
    from miccai2021_cataract_semantic_segmentation_amd.models import EncDec
    model = EncDec({"encoder": {"model": "ResNeXt101", "pretrained": False}, "decoder": {"model": "UPerNet"}}, 3)
    model.get_features = False
    spec = spec_of(model.state_dict())
    S = fill_state(spec, 47)
    model.load_state_dict(S)
    model.cuda().eval()
    g = torch.Generator().manual_seed(13)
    x = torch.rand(1, 3, 1088, 1920, generator=g)
    with torch.no_grad():
        out = model(x.cuda())
        out = out[0] if isinstance(out, (tuple, list)) else out
        ref = upernet_forward(S, _resnext_oracle(S, x), False)
        ref = ref[0] if isinstance(ref, (tuple, list)) else ref
        S64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in S.items()}
        f64 = upernet_forward(S64, _resnext_oracle(S64, x.double()), False)
        f64 = f64[0] if isinstance(f64, (tuple, list)) else f64
    out_h = out.detach().cpu()
    e_abs, scale = float((out_h - ref).abs().max()), float(ref.abs().max())
    print("ResNeXt101-UPerNet 1x1088x1920: max |logit - cpu32| %.3g (relative %.3g of scale %.3g)" % (e_abs, e_abs / scale, scale))
    assert e_abs <= 1e-3 * max(1.0, scale)
    _argmax_report("ResNeXt101-UPerNet full resolution", out_h, ref, f64)
