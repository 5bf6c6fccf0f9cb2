"""Functional fp32 CPU restatement of the reference networks (oracle; test infrastructure only).

All functions take ``S``: an ordered dict  name -> tensor  using the reference's
state_dict key names (``backbone.layer3.0.conv1.weight`` ...), so one state dict
drives the reference, this oracle and the HIP product path alike.  BatchNorm
running statistics in ``S`` are updated in place in training mode exactly as
``torch.nn.BatchNorm2d`` does (momentum 0.1, unbiased variance for the running
estimate, biased for normalisation).
"""
import torch
import torch.nn.functional as F

TAPS = None   # diagnostic hook (tools/error_growth.py): a dict collects named intermediate activations


def tap(name, t):
    if TAPS is not None:
        TAPS[name] = t.detach().clone()
    return t


def bn(S, p, x, train, eps=1e-5, momentum=0.1):
    if train and (p + ".num_batches_tracked") in S:
        S[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, S[p + ".running_mean"], S[p + ".running_var"], S[p + ".weight"],
                        S[p + ".bias"], train, momentum, eps)


def conv(S, p, x, stride=1, padding=0, dilation=1):
    return F.conv2d(x, S[p + ".weight"], S.get(p + ".bias"), stride, padding, dilation)


# ----------------------------------------------------------------------------- ResNet (torchvision v1.5)
# PARITY UNPINNED (third-party torchvision; see oracle/resnet_tv.py docstring).
# call sites: models/OCR.py:58-61, models/DeepLabv3Plus.py:35-41
RESNET_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


def resnet_plan(name, replace_stride_with_dilation):
    """[(layer, block, inplanes, planes, stride, dilation, has_downsample)] for Bottleneck ResNets."""
    plan, inpl, dil = [], 64, 1
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], RESNET_LAYERS[name])):
        stride = 1 if li == 0 else 2
        prev = dil
        if li > 0 and replace_stride_with_dilation[li - 1]:
            dil *= stride
            stride = 1
        for b in range(nblk):
            s = stride if b == 0 else 1
            d = prev if b == 0 else dil
            plan.append((li + 1, b, inpl, planes, s, d, b == 0 and (s != 1 or inpl != planes * 4)))
            inpl = planes * 4
    return plan


def bottleneck(S, p, x, stride, dilation, has_ds, train):
    idt = x
    if has_ds:
        idt = bn(S, p + ".downsample.1", conv(S, p + ".downsample.0", x, stride), train)
    o = F.relu(bn(S, p + ".bn1", conv(S, p + ".conv1", x), train))
    o = F.relu(bn(S, p + ".bn2", conv(S, p + ".conv2", o, stride, dilation, dilation), train))
    o = bn(S, p + ".bn3", conv(S, p + ".conv3", o), train)
    return F.relu(o + idt)


def resnet_features(S, x, name, rswd, train, prefix="backbone."):
    """Returns {1: layer1 out, ..., 4: layer4 out}."""
    x = F.relu(bn(S, prefix + "bn1", conv(S, prefix + "conv1", x, 2, 3), train))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = {}
    for (li, b, _, _, s, d, ds) in resnet_plan(name, rswd):
        x = bottleneck(S, "%slayer%d.%d" % (prefix, li, b), x, s, d, ds, train)
        feats[li] = x
    return feats


def rswd_for(out_stride):
    # models/OCR.py:49-56 ; DeepLabv3Plus.py:19-25 (identical for 8 and 16)
    return {8: [False, True, True], 16: [False, False, True]}[out_stride]


# ----------------------------------------------------------------------------- OCR head
def spatial_gather(feats, logits):
    """models/OCR.py:158-170  feats B,C,H,W ; logits B,K,H,W -> B,C,K,1"""
    B, K = logits.shape[:2]
    probs = F.softmax(logits.reshape(B, K, -1), dim=2)
    f = feats.reshape(B, feats.shape[1], -1).permute(0, 2, 1)
    return torch.matmul(probs, f).permute(0, 2, 1).unsqueeze(3)


def _cbr(S, p, i, x, train):  # Sequential(conv i, bn i+1, relu)
    return F.relu(bn(S, "%s.%d" % (p, i + 1), conv(S, "%s.%d" % (p, i), x), train))


def object_attention(S, p, x, proxy, train, key_channels=256):
    """models/OCR.py:237-284 (scale == 1)"""
    B, _, H, W = x.shape
    tap("proxy", proxy)
    q = tap("q", _cbr(S, p + ".f_pixel", 3, _cbr(S, p + ".f_pixel", 0, x, train), train))
    q = q.reshape(B, key_channels, -1).permute(0, 2, 1)
    k = tap("k", _cbr(S, p + ".f_object", 3, _cbr(S, p + ".f_object", 0, proxy, train), train)).reshape(B, key_channels, -1)
    v = tap("v", _cbr(S, p + ".f_down", 0, proxy, train)).reshape(B, key_channels, -1).permute(0, 2, 1)
    sim = F.softmax((key_channels ** -.5) * torch.matmul(q, k), dim=-1)
    ctx = tap("ctx", torch.matmul(sim, v).permute(0, 2, 1).contiguous().reshape(B, key_channels, H, W))
    return tap("context", _cbr(S, p + ".f_up", 0, ctx, train))


def ocrnet_forward(S, x, backbone="resnet50", out_stride=8, train=True):
    """models/OCR.py:107-138 -> (interm_up_logits, up_logits)"""
    size = x.shape[-2:]
    f = resnet_features(S, x, backbone, rswd_for(out_stride), train)
    low, high = f[3], f[4]
    h = F.relu(bn(S, "interm_prediction_head.1", conv(S, "interm_prediction_head.0", low, 1, 1), train))
    interm = conv(S, "interm_prediction_head.4", h)                 # Dropout2d(p=0) is the identity
    xh = F.relu(bn(S, "conv_high_map.1", conv(S, "conv_high_map.0", high, 1, 1), train))
    proxy = spatial_gather(xh, interm)
    ctx = object_attention(S, "spatial_ocr_head.object_context_block", xh, proxy, train)
    o = torch.cat([ctx, xh], 1)                                       # models/OCR.py:320
    o = F.relu(bn(S, "spatial_ocr_head.conv_bn_dropout.1", conv(S, "spatial_ocr_head.conv_bn_dropout.0", o), train))
    logits = conv(S, "conv_out", o)
    up = F.interpolate(logits, size=size, mode="bilinear", align_corners=True)
    iup = F.interpolate(interm, size=size, mode="bilinear", align_corners=True)
    return iup, up


# ----------------------------------------------------------------------------- DeepLabv3+
def aspp(S, p, x, mult, train):
    """models/DeepLabv3Plus.py:106-129 ; BN eps = 3e-4 (the positional 'momentum' lands in eps, :85,98-104)"""
    e = 3e-4
    outs = [F.relu(bn(S, p + "aspp1_bn", conv(S, p + "aspp1", x), train, e))]
    for i, r in ((2, 6), (3, 12), (4, 18)):
        d = int(r * mult)
        outs.append(F.relu(bn(S, p + "aspp%d_bn" % i, conv(S, p + "aspp%d" % i, x, 1, d, d), train, e)))
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(bn(S, p + "aspp5_bn", conv(S, p + "aspp5", g), train, e))
    outs.append(F.interpolate(g, size=x.shape[2:], mode="bilinear", align_corners=True))
    y = torch.cat(outs, 1)
    return F.relu(bn(S, p + "bn2", conv(S, p + "conv2", y), train, e))


def deeplab_decoder(S, p, low, a, train):
    """models/DeepLabv3Plus.py:158-175"""
    e = 3e-4
    x1 = F.relu(bn(S, p + "conv_low_bn", conv(S, p + "conv_low", low), train, e))
    x2 = F.interpolate(a, size=low.shape[2:], mode="bilinear", align_corners=True)
    y = torch.cat((x1, x2), 1)
    y = F.relu(bn(S, p + "conv_3x3_1_bn", conv(S, p + "conv_3x3_1", y, 1, 1), train, e))
    y = F.relu(bn(S, p + "conv_3x3_2_bn", conv(S, p + "conv_3x3_2", y, 1, 1), train, e))
    return conv(S, p + "conv_out", y)


def deeplabv3plus_forward(S, x, backbone="resnet50", out_stride=8, train=True):
    """models/DeepLabv3Plus.py:58-74"""
    size = x.shape[-2:]
    f = resnet_features(S, x, backbone, rswd_for(out_stride), train)
    a = aspp(S, "aspp.", f[4], 1 if out_stride >= 16 else 2, train)
    logits = deeplab_decoder(S, "decoder.", f[1], a, train)
    return F.interpolate(logits, size=size, mode="bilinear", align_corners=True)


def ocrnet_hrnet_forward(S, x, train=True):
    """Build-side assembly (the reference has no HRNet OCRNet, models/OCR.py:68-69): HRNetv2 trunk
    (oracle/hrnet.py, keys under 'backbone.') -> stride-4 concat -> the reference's OCR heads
    (models/OCR.py:107-138) with low = high = the concat."""
    from .hrnet import hrnet_branches, hrnet_concat
    size = x.shape[-2:]
    f = tap("concat", hrnet_concat(hrnet_branches(S, x, train, prefix="backbone.")))
    h = F.relu(bn(S, "interm_prediction_head.1", conv(S, "interm_prediction_head.0", f, 1, 1), train))
    interm = tap("interm_lowres", conv(S, "interm_prediction_head.4", h))
    xh = tap("feats", F.relu(bn(S, "conv_high_map.1", conv(S, "conv_high_map.0", f, 1, 1), train)))
    proxy = spatial_gather(xh, interm)
    ctx = object_attention(S, "spatial_ocr_head.object_context_block", xh, proxy, train)
    o = torch.cat([ctx, xh], 1)
    o = tap("ocr_out", F.relu(bn(S, "spatial_ocr_head.conv_bn_dropout.1", conv(S, "spatial_ocr_head.conv_bn_dropout.0", o), train)))
    logits = tap("logits_lowres", conv(S, "conv_out", o))
    return (F.interpolate(interm, size=size, mode="bilinear", align_corners=True),
            F.interpolate(logits, size=size, mode="bilinear", align_corners=True))


def deeplabv3_forward(S, x, backbone="resnet50", out_stride=8, train=True):
    """models/DeepLabv3.py:58-71"""
    size = x.shape[-2:]
    f = resnet_features(S, x, backbone, rswd_for(out_stride), train)
    logits = conv(S, "conv_out", aspp(S, "aspp.", f[4], 1 if out_stride >= 16 else 2, train))
    return F.interpolate(logits, size=size, mode="bilinear", align_corners=True)


def fcn_forward(S, x):
    """models/FCN.py:40-61 of the reference (FCN-8s; no normalisation layers, so training and inference run the same arithmetic).
    Paddings from utils/torch_utils.py:130-168: 3x3 -> 1, 1x1 -> 0; ConvTranspose2d 4 / stride 2 -> 1, 16 / stride 8 -> 4."""
    def deconv(p, t, stride, pad):
        return F.conv_transpose2d(t, S[p + ".weight"], S.get(p + ".bias"), stride, pad)

    p1 = F.max_pool2d(F.relu(conv(S, "conv1", x, 1, 1)), 2)
    p2 = F.max_pool2d(F.relu(conv(S, "conv2", p1, 1, 1)), 2)
    p3 = F.max_pool2d(F.relu(conv(S, "conv3", p2, 1, 1)), 2)
    p4 = F.max_pool2d(F.relu(conv(S, "conv4", p3, 1, 1)), 2)
    p5 = F.max_pool2d(F.relu(conv(S, "conv5", p4, 1, 1)), 2)
    c7 = F.relu(conv(S, "conv7", F.relu(conv(S, "conv6", p5, 1, 1))))
    fcn_16s = deconv("deconv32", conv(S, "conv8", c7), 2, 1) + conv(S, "p4_conv", p4)
    fcn_8s = deconv("deconv16", fcn_16s, 2, 1) + conv(S, "p3_conv", p3)
    return deconv("deconv8", fcn_8s, 8, 4)
