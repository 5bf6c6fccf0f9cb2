"""Full-resolution parity under the PRODUCTION plan: the networks at the bench sizes (3 x 544 x 960, config 5 at 3 x 1088 x 1920), the
default arithmetic selection (ops.PRECISION / TRUNK / HEADS as shipped: two fp16 planes on the HRNet trunk's direct kernels and on the
head layers, exact fp32 elsewhere), NO threshold forcing -- the tile forms, blocked planes, split counts and persistent-block schedules
of the timed benchmark -- against the CPU oracle evaluated at the same size (one oracle evaluation per network, shared by the assertions
and by every arithmetic plan of tests/_fullres.py: production, fp32, trunk_bf16x3, heads_bf16x3, all_bf16x3).

The oracle is test infrastructure (oracle/__init__.py); every HIP call goes through the C ABI.  THE BAR (asserted by `_logit_bar`, every
figure also written to the JSON record gpurun_out/parity_fullres.json so that a passing run leaves evidence; a copy is kept as
profiles/r04_parity_fullres.json):
  (a) max |logit_hip - logit_fp64| <= 1e-3 ABSOLUTE against the fp64 evaluation of the oracle (the north star's 1e-3);
  (b) the same distance <= 1.2 x the fp32 CPU oracle's own distance to fp64 (+1e-5): the HIP path is not noisier than the reference
      arithmetic it replaces (production plan; the other plans: <= 2 x);
  (c) max |logit_hip - logit_cpu32| <= 1e-3 of the logit scale (two fp32 evaluations that are each ~0.7e-3 absolute from fp64 can be
      1.4e-3 apart: an absolute 1e-3 between THEM is not a property either has; the absolute figure is recorded);
  (d) label maps: no disagreement with the fp64 map where the fp64 top-2 margin exceeds 2.2 x the logit error made, and the UNMASKED
      count of disagreements with fp64 <= 1.25 x the fp32 CPU run's own count (+8) in the production plan.
Loss 1e-4, BatchNorm running statistics 1e-4 relative, per-tensor gradient norms and directions against the fp32 CPU oracle on a fixed
subset."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


from _fullres import block_labels as _labels  # noqa: E402
import _fullres as FR  # noqa: E402


def _logit_bar(section, plan, hip, cpu32, f64, production=True, extra=None, noise_factor=None):
    """bars (a) - (d) of the module docstring; records the figures.  (a) is absolute for logits of the trained-like scale (<= 10); a
    random-weight network whose logits explode (eval-mode ResNeXt101-UPerNet: 4e5) is held to 1e-4 of its scale instead"""
    scale = float(cpu32.abs().max())
    fig = {"logit_scale": scale, "e_abs_vs_cpu32": float((hip - cpu32).abs().max()), "e_abs_vs_fp64": float((hip.double() - f64).abs().max()),
           "cpu32_e_abs_vs_fp64": float((cpu32.double() - f64).abs().max()),
           "e_rms_vs_fp64": float((hip.double() - f64).pow(2).mean().sqrt()), "cpu32_e_rms_vs_fp64": float((cpu32.double() - f64).pow(2).mean().sqrt()),
           "argmax": FR.argmax_figures(hip, cpu32, f64)}
    fig.update(extra or {})
    FR.record(section, plan, fig)
    print(section, plan, fig)
    am = fig["argmax"]
    assert fig["e_abs_vs_fp64"] <= 1e-3 * max(1.0, scale / 10), ("(a) absolute distance to the fp64 oracle", fig)
    nf = noise_factor or (1.2 if production else 2.0)
    assert fig["e_abs_vs_fp64"] <= nf * fig["cpu32_e_abs_vs_fp64"] + 1e-5, ("(b) noisier than the fp32 CPU path", fig)
    assert fig["e_abs_vs_cpu32"] <= 1e-3 * max(1.0, scale), ("(c) distance to the fp32 CPU oracle", fig)
    assert am["outside_error_band"] == 0, ("(d) label differs from the fp64 oracle where the margin exceeds the logit error", fig)
    if production:
        assert am["hip_vs_fp64"] <= 1.25 * am["cpu32_vs_fp64"] + 8, ("(d) unmasked label-map disagreements", fig)
    else:
        assert am["hip_vs_fp64"] <= 2 * am["cpu32_vs_fp64"] + 32, ("(d) unmasked label-map disagreements", fig)
    return fig


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


@pytest.mark.parametrize("plan", list(FR.PLANS))
def test_ocrnet_hrnet48_fullres_train_step_vs_oracle(plan):
    """the BENCH model at the BENCH resolution (batch 2 of 8: the CPU oracle's fp64 forward has to fit the test budget) under every
    arithmetic plan (ONE oracle evaluation for all of them): logits and label maps for each; loss, BatchNorm statistics, gradients and
    the kernel populations for the production plan"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    assert ops.PRECISION == "bf16x3" and ops.TRUNK == "f16x2" and ops.HEADS == "f16x2" and ops.PLANES, "shipped defaults expected"
    assert ops.B3_MIN_K == 2048 and ops.DCONV3_MIN_ROWS == 2048, "production plan expected"
    orc = FR.hrnet48_oracle()
    model, interm_h, final_h, loss_h, kinds = FR.hrnet48_hip(orc, plan)
    S = orc["S"]
    production = plan == "production"
    fig = _logit_bar("ocrnet_hrnet48_2x544x960", plan, final_h, orc["final32"], orc["final64"], production,
                     {"e_abs_interm_vs_cpu32": float((interm_h - orc["interm32"]).abs().max()),
                      "e_abs_interm_vs_fp64": float((interm_h.double() - orc["interm64"]).abs().max()),
                      "cpu32_e_abs_interm_vs_fp64": float((orc["interm32"].double() - orc["interm64"]).abs().max()),
                      "loss_hip": loss_h, "loss_cpu32": orc["loss32"], "loss_fp64": orc["loss64"],
                      "kernel_populations": sorted(k for k in kinds if not k.startswith("hbm:"))})
    assert fig["e_abs_interm_vs_cpu32"] <= 1e-3 * max(1.0, float(orc["interm32"].abs().max()))
    assert abs(loss_h - orc["loss32"]) < 1e-4
    # the kernels of the plan really ran
    P, T, Hd, PL = FR.PLANS[plan]
    if P == "fp32":
        assert {"fwd", "dgrad", "wgrad"} <= kinds and not any("_d3" in k or "_h2" in k or "_b3" in k for k in kinds), kinds
    else:
        heads = {"fwd_h2", "dgrad_h2", "wgrad_h2"} if Hd == "f16x2" else {"fwd_b3", "dgrad_b3", "wgrad_b3"}
        trunk = ({"fwd_d3p", "dgrad_d3p", "wgrad_d3p"} if PL else {"fwd_d3h", "dgrad_d3h", "wgrad_d3h"}) if T == "f16x2" else {"fwd_d3", "dgrad_d3", "wgrad_d3"}
        assert heads | trunk | {"fwd", "dgrad", "wgrad"} <= kinds, kinds
        if PL and T == "f16x2":     # the planes route took every trunk layer but the first convolution behind each transition (its input has no record)
            assert not any(k.endswith("_d3") for k in kinds), kinds
        if T == "f16x2" and ops.P1 and ops.G1:
            # round 5: the pointwise and gather launches of csrc/pconv1.hip took the layers the benchmark runs them on (the pixel-count
            # thresholds scaled to this batch: _fullres.set_plan)
            assert {"fwd_p1", "dgrad_p1", "wgrad_p1", "fwd_s2p", "dgrad_s2p", "wgrad_s2p"} <= kinds, kinds
    if not production:
        return
    sd = model.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.stage2.0.branches.0.1.bn1.running_var", "backbone.stage4.2.branches.3.3.bn2.running_var",
              "backbone.stage3.1.branches.1.2.bn2.running_mean", "conv_high_map.1.running_var"):
        a, b = sd[k].detach().cpu().double(), S[k].detach().double()
        assert float((a - b).abs().max()) <= 1e-5 + 1e-4 * float(b.abs().max()), k
    keys = ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight", "backbone.stage2.0.branches.0.0.conv1.weight",
            "backbone.stage2.0.branches.1.3.conv2.weight", "backbone.stage3.2.branches.2.1.conv1.weight",
            "backbone.stage4.1.branches.3.2.conv2.weight", "backbone.stage4.2.branches.0.3.conv1.weight",
            "backbone.stage3.0.fuse_layers.2.0.0.0.weight", "backbone.stage4.0.fuse_layers.0.3.0.weight",
            "backbone.stage2.0.branches.0.2.bn1.weight", "conv_high_map.0.weight", "interm_prediction_head.3.weight",
            "ocr_distri_head.object_context_block.f_pixel.0.weight", "final_prediction_head.weight",
            # round 6: both class heads run fused with the BatchNorm in front of them (csrc/headfuse.h): the classifier, the BatchNorm and the
            # convolution in front of it, each against the oracle's autograd
            "interm_prediction_head.4.weight", "interm_prediction_head.4.bias", "interm_prediction_head.1.weight", "interm_prediction_head.1.bias",
            "interm_prediction_head.0.weight", "conv_out.weight", "conv_out.bias", "spatial_ocr_head.conv_bn_dropout.1.weight",
            "spatial_ocr_head.conv_bn_dropout.1.bias", "spatial_ocr_head.conv_bn_dropout.0.weight",
            "spatial_ocr_head.object_context_block.f_up.0.weight"]
    if ops.HEAD_FUSE:
        assert {"hbm:head_fwd", "hbm:head_backward"} <= kinds, "the class heads of the bench model did not take the fused route"
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
    _logit_bar("deeplabv3plus_r50_2x544x960", "production", out_h, ref_d, f64, True, {"loss_hip": float(loss), "loss_cpu32": float(ol)})
    assert abs(float(loss) - float(ol)) < 1e-4
    have = dict(model.named_parameters())
    keys = [k for k in ("backbone.conv1.weight", "backbone.layer2.1.conv2.weight", "backbone.layer4.2.conv2.weight", "aspp.convs.1.0.weight",
                        "aspp.convs.3.0.weight", "aspp.project.0.weight", "decoder.conv_low.0.weight", "decoder.conv_out.0.weight",
                        "decoder.conv_out.6.weight") if k in have and k in S]
    if keys:
        _grad_subset_check("DeepLabv3+-R50 full resolution", model, S, keys)


def test_resnext101_upernet_fullres_inference_vs_oracle():
    """BASELINE config 5 at its resolution and at its per-GPU batch (4 frames of 3 x 1088 x 1920 = bs 32 over 8 GPUs, reference
    models/UPerNet.py:108-145, managers/BaseManager.py:640-688): the fused inference path (folded BatchNorm, bf16x3 blocked kernels on
    conv_last / FPN) -- every frame of the four-frame call bit-identical to its single-frame run, frame 0 against the CPU restatement"""
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
    # configuration 5 as stated: bs 32 sharded by frame over 8 GPUs = FOUR frames of 3 x 1088 x 1920 per GPU in one call.  Eval-mode
    # BatchNorm makes a frame's logits mathematically independent of its batch mates; the kernels' tile / split plans are chosen from the
    # row count of a call (4 frames: other tile forms and batch pieces than 1 frame), so the two calls round differently: each frame of
    # the four-frame call must agree with its single-frame run to 2e-5 of the logit scale (measured 4.5e-6) with label maps that differ
    # only where the top-2 margin is inside that distance, and frame 0 of the four-frame call is held to the oracle bars below as well
    x4 = torch.cat([x, torch.rand(3, 3, 1088, 1920, generator=g)]).cuda()
    with torch.no_grad():
        out4 = model(x4)
        out4 = (out4[0] if isinstance(out4, (tuple, list)) else out4).detach()
        assert out4.shape[0] == 4
        for i in range(4):
            oi = model(x4[i:i + 1])
            oi = (oi[0] if isinstance(oi, (tuple, list)) else oi).detach()
            o4 = out4[i:i + 1]
            d, sc = float((o4 - oi).abs().max()), float(oi.abs().max())
            top2 = oi.topk(2, dim=1).values
            differ = o4.argmax(1) != oi.argmax(1)
            outside = int((differ & ((top2[:, 0] - top2[:, 1]) > 2.2 * d)).sum())
            FR.record("resnext101_upernet_4x1088x1920_inference", "frame%d_vs_single_frame_run" % i,
                      {"max_abs_diff": d, "logit_scale": sc, "bit_identical": bool(torch.equal(o4, oi)), "labels_differ": int(differ.sum()),
                       "labels_differ_outside_error_band": outside, "pixels": int(differ.numel())})
            assert d <= 2e-5 * sc and outside == 0, (i, d, sc, outside)
        out4_0 = out4[0:1].cpu()
        del out4, oi, o4, x4
        out = model(x.cuda())
        out = out[0] if isinstance(out, (tuple, list)) else out
        ref = upernet_forward(S, _resnext_oracle(S, x), False)
        ref = ref[0] if isinstance(ref, (tuple, list)) else ref
        S64 = {k: (v.double() if v.dtype.is_floating_point else v.clone()) for k, v in S.items()}
        f64 = upernet_forward(S64, _resnext_oracle(S64, x.double()), False)
        f64 = f64[0] if isinstance(f64, (tuple, list)) else f64
    out_h = out.detach().cpu()
    # (b) with factor 4: the fused inference path folds BatchNorm into the weights (one more rounding of every weight) and runs the wide
    # layers on three bf16 planes; measured 3.5e-6 of the logit scale against the fp32 CPU oracle's 1.1e-6, label maps identical
    _logit_bar("resnext101_upernet_1x1088x1920_inference", "production", out_h, ref, f64, True, noise_factor=4.0)
    # (the four-frame call cuts its batch into other pieces / tile forms: measured 5.1e-6 of the logit scale from fp64 against 3.5e-6 for the
    #  one-frame call and 1.1e-6 for the fp32 CPU oracle; label maps identical to fp64 in all three)
    _logit_bar("resnext101_upernet_4x1088x1920_inference", "frame0_of_four_vs_oracle", out4_0, ref, f64, True, noise_factor=6.0)


def test_config3_batch8_production_vs_exact_fp32_cross_plan():
    """configuration 3 at its REAL batch (8 x 3 x 544 x 960, TwoScale-Lovasz, the bench model): the production arithmetic against an
    INDEPENDENT one -- CATSEG_PRECISION=fp32, exact fp32 MFMA chains in every layer -- on the HIP path itself (no CPU oracle can be afforded
    at this size: ~3 minutes and 30 GB per evaluation).  Both plans are held to 1e-3 absolute of fp64 at batch 2 by the tests above; at
    batch 8 they must agree with each other to the sum of those bars, give the same loss, the same label maps up to ties, and gradients of
    the same size and direction.  The production step is then replayed as a hipGraph and must reproduce its own bits."""
    _need_gpu()
    import bench
    from oracle.state import fill_state, spec_of
    from miccai2021_cataract_semantic_segmentation_amd import ops
    from miccai2021_cataract_semantic_segmentation_amd.graph import GraphedTrainStep
    from miccai2021_cataract_semantic_segmentation_amd.losses import TwoScaleLoss
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.optim import FusedAdam
    cfg = dict(bench.MODELS["ocrnet_hrnet48"][0])
    spec = spec_of(OCRNet(dict(cfg), 3).state_dict())
    x, lbl = bench.synth_batch(8, 544, 960, 25, 77, torch.device("cuda"))
    crit = TwoScaleLoss({"experiment": 3, "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                         "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}})
    keys = ["backbone.conv1.weight", "backbone.layer1.0.conv1.weight", "backbone.stage2.0.branches.1.0.conv1.weight",
            "backbone.stage3.1.fuse_layers.2.0.0.0.weight", "backbone.stage4.2.branches.3.3.conv2.weight", "conv_high_map.0.weight",
            "interm_prediction_head.0.weight", "spatial_ocr_head.object_context_block.f_pixel.0.weight",
            "spatial_ocr_head.object_context_block.f_up.0.weight", "spatial_ocr_head.conv_bn_dropout.0.weight", "conv_out.weight"]
    res = {}
    for plan in ("production", "fp32"):
        with FR.set_plan(plan):
            model = OCRNet(dict(cfg), 3)
            model.load_state_dict(fill_state(spec, 41))
            model.cuda().train()
            interm, final = model(x)
            loss = crit(interm, final, lbl)
            loss.backward()
            torch.cuda.synchronize()
            named = dict(model.named_parameters())
            res[plan] = (final.detach().float().clone(), float(loss.detach()), {k: named[k].grad.detach().clone() for k in keys if k in named})
            if plan == "production":
                # the same step through the graph: identical logits, loss and flat gradient, bit for bit
                g_eager = model.flat().grad.clone()
                opt = FusedAdam(model, lr=0.0)
                step = GraphedTrainStep(model, lambda o, l: crit(*o, l), opt, x, lbl)
                l_g = float(step(x, lbl))
                torch.cuda.synchronize()
                assert l_g == res[plan][1] and torch.equal(step.outputs[1].detach(), res[plan][0]) and torch.equal(model.flat().grad, g_eager)
                step.release()
            del model, interm, final, loss
            torch.cuda.empty_cache()
    (fp, lp, gp), (f3, l3, g3) = res["production"], res["fp32"]
    d = float((fp - f3).abs().max())
    scale = float(f3.abs().max())
    lab = int((fp.argmax(1) != f3.argmax(1)).sum())
    top2 = f3.topk(2, dim=1).values
    outside = int(((fp.argmax(1) != f3.argmax(1)) & ((top2[:, 0] - top2[:, 1]) > 2.2 * d)).sum())
    fig = {"max_abs_logit_difference": d, "logit_scale": scale, "loss_production": lp, "loss_exact_fp32": l3, "label_disagreements": lab,
           "pixels": int(fp[:, 0].numel()), "label_disagreements_outside_the_error_band": outside}
    rows = []
    for k in gp:
        a, b = gp[k].double().reshape(-1), g3[k].double().reshape(-1)
        rows.append((k, float(a.norm() / (b.norm() + 1e-300)), float((a * b).sum() / (a.norm() * b.norm() + 1e-300))))
    fig["gradients"] = {k: {"norm_ratio": r, "cosine": c} for k, r, c in rows}
    FR.record("ocrnet_hrnet48_8x544x960", "production_vs_exact_fp32", fig)
    print(fig)
    assert len(rows) >= 8
    assert d <= 2e-3, fig                                  # each plan within 1e-3 absolute of fp64 (asserted at batch 2 above)
    assert abs(lp - l3) <= 1e-5 * max(1.0, abs(l3)), fig
    assert outside == 0 and lab <= 4e-4 * fig["pixels"], fig
    for k, r, c in rows:
        assert abs(r - 1) < 2e-2 and c > 0.999, (k, r, c)
