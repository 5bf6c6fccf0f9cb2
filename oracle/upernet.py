"""Functional fp32 CPU restatement of the reference's EncDec + UPerNet (models/EncDec.py:43-53,
models/UPerNet.py:108-145) with the torchvision ResNet18/34 encoder wrappers (models/ResNet.py:9-26);
oracle, test infrastructure only.  PARITY UNPINNED for the torchvision trunk (see oracle/resnet_tv.py)."""
import torch
import torch.nn.functional as F

from .nets import bn, conv

BASIC_LAYERS = {"ResNet18": [2, 2, 2, 2], "ResNet34": [3, 4, 6, 3]}


def basic_block(S, p, x, stride, train):
    idt = x
    if (p + ".downsample.0.weight") in S:
        idt = bn(S, p + ".downsample.1", conv(S, p + ".downsample.0", x, stride), train)
    o = F.relu(bn(S, p + ".bn1", conv(S, p + ".conv1", x, stride, 1), train))
    o = bn(S, p + ".bn2", conv(S, p + ".conv2", o, 1, 1), train)
    return F.relu(o + idt)


def resnet_basic_stages(S, x, name, train, prefix="enc_model."):
    """models/ResNet.py:9-26: returns [layer1, layer2, layer3, layer4] outputs"""
    x = F.relu(bn(S, prefix + "bn1", conv(S, prefix + "conv1", x, 2, 3), train))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for li, n in enumerate(BASIC_LAYERS[name]):
        for b in range(n):
            x = basic_block(S, "%slayer%d.%d" % (prefix, li + 1, b), x, 2 if (li > 0 and b == 0) else 1, train)
        outs.append(x)
    return outs


def _cbr(S, p, x, train, k3=False):
    return F.relu(bn(S, p + ".1", conv(S, p + ".0", x, 1, 1 if k3 else 0), train))


def upernet_forward(S, conv_out, train, prefix="dec_model.", pool_scales=(1, 2, 3, 6), in_scale=4):
    """models/UPerNet.py:108-145"""
    P = prefix
    conv5 = conv_out[-1]
    size = conv5.shape[2:]
    ppm = [conv5]
    for k, s in enumerate(pool_scales):
        t = F.interpolate(F.adaptive_avg_pool2d(conv5, s), size, mode="bilinear", align_corners=False)
        ppm.append(_cbr(S, "%sppm_conv.%d" % (P, k), t, train))
    feature = _cbr(S, P + "ppm_last_conv", torch.cat(ppm, 1), train, k3=True)
    fpn = [feature]
    L = len(conv_out)
    for i in range(2, L + 1):
        cx = _cbr(S, "%sfpn_in.%d" % (P, L - i), conv_out[-i], train)          # fpn_in[-i+1] of an (L-1)-long list
        feature = cx + F.interpolate(feature, size=cx.shape[2:], mode="bilinear", align_corners=False)
        fpn.append(_cbr(S, "%sfpn_out.%d.0" % (P, L - i), feature, train, k3=True))
    fpn.reverse()
    osz = fpn[0].shape[2:]
    fus = [fpn[0]] + [F.interpolate(fpn[-i + 1], osz, mode="bilinear", align_corners=False) for i in range(2, L + 1)]
    x = _cbr(S, P + "conv_last.0", torch.cat(fus, 1), train, k3=True)
    x = conv(S, P + "conv_last.1", x)
    return F.interpolate(x, scale_factor=in_scale, mode="bilinear", align_corners=False)


def encdec_forward(S, x, encoder="ResNet18", train=True):
    """models/EncDec.py:43-53 -> (deep_features, prediction)"""
    feats = resnet_basic_stages(S, x, encoder, train)
    return feats[-1], upernet_forward(S, feats, train)


def resnext_features_eval(S, x, layers=(3, 4, 23, 3), groups=32, prefix="enc_model."):
    """torchvision resnext101_32x8d trunk in eval mode behind the reference's wrapper (models/ResNeXt.py:46-60: the four stage outputs at
    strides 4 / 8 / 16 / 32): Bottleneck v1.5 with the stride on the grouped 3x3, expansion 4, 1x1 + BatchNorm downsample.  torchvision is
    an unpinned, un-vendored dependency of the reference: parity unpinned for this trunk."""
    def bn(p, t):
        return F.batch_norm(t, S[p + ".running_mean"], S[p + ".running_var"], S[p + ".weight"], S[p + ".bias"], False, 0.1, 1e-5)
    t = F.max_pool2d(F.relu(bn(prefix + "bn1", F.conv2d(x, S[prefix + "conv1.weight"], None, 2, 3))), 3, 2, 1)
    outs = []
    for li, n in enumerate(layers):
        for b in range(n):
            p = "%slayer%d.%d" % (prefix, li + 1, b)
            s = 2 if (li > 0 and b == 0) else 1
            idt = t
            if (p + ".downsample.0.weight") in S:
                idt = bn(p + ".downsample.1", F.conv2d(t, S[p + ".downsample.0.weight"], None, s))
            o = F.relu(bn(p + ".bn1", F.conv2d(t, S[p + ".conv1.weight"])))
            o = F.relu(bn(p + ".bn2", F.conv2d(o, S[p + ".conv2.weight"], None, s, 1, 1, groups)))
            o = bn(p + ".bn3", F.conv2d(o, S[p + ".conv3.weight"]))
            t = F.relu(o + idt)
        outs.append(t)
    return outs


def resnext101_upernet_infer(S, x):
    """EncDec(ResNeXt101_32x8d + UPerNet) eval-mode forward (BASELINE config 5; reference models/EncDec.py:43-53, UPerNet.py:108-145)"""
    out = upernet_forward(S, resnext_features_eval(S, x), False)
    return out[0] if isinstance(out, (tuple, list)) else out
