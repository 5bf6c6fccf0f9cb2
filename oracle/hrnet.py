"""Functional fp32 CPU restatement of the reference's HRNetv2 (models/HRNetv2.py:36-514); oracle, test
infrastructure only.  ``S`` uses the reference's state-dict keys.  BatchNorm momentum 0.01 (:19); every
bilinear resize uses align_corners=False (the F.interpolate default, :253-256,504-512)."""
import torch
import torch.nn.functional as F

from .nets import bn as _bn_generic, conv, tap

MOM = 0.01


def bn(S, p, x, train):
    return _bn_generic(S, p, x, train, 1e-5, MOM)


def basic_block(S, p, x, train):
    """:36-65"""
    o = F.relu(bn(S, p + ".bn1", conv(S, p + ".conv1", x, 1, 1), train))
    o = bn(S, p + ".bn2", conv(S, p + ".conv2", o, 1, 1), train)
    return F.relu(o + x)


def bottleneck(S, p, x, train):
    """:68-106 (stride 1; only the first block of layer1 has a downsample)"""
    idt = x
    if (p + ".downsample.0.weight") in S:
        idt = bn(S, p + ".downsample.1", conv(S, p + ".downsample.0", x), train)
    o = F.relu(bn(S, p + ".bn1", conv(S, p + ".conv1", x), train))
    o = F.relu(bn(S, p + ".bn2", conv(S, p + ".conv2", o, 1, 1), train))
    o = bn(S, p + ".bn3", conv(S, p + ".conv3", o), train)
    return F.relu(o + idt)


def _count(S, prefix):
    """number of consecutive integer children prefix.0, prefix.1, ... present in S"""
    n = 0
    while any(k.startswith("%s.%d." % (prefix, n)) for k in S):
        n += 1
    return n


def hr_module(S, p, xs, train):
    """HighResolutionModule.forward :237-261"""
    nb = len(xs)
    for i in range(nb):
        for b in range(_count(S, "%s.branches.%d" % (p, i))):
            xs[i] = basic_block(S, "%s.branches.%d.%d" % (p, i, b), xs[i], train)
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            f = "%s.fuse_layers.%d.%d" % (p, i, j)
            if j == i:
                t = xs[j]
            elif j > i:
                t = bn(S, f + ".1", conv(S, f + ".0", xs[j]), train)
                t = F.interpolate(t, size=xs[i].shape[-2:], mode="bilinear")
            else:
                t = xs[j]
                for k in range(i - j):
                    t = bn(S, "%s.%d.1" % (f, k), conv(S, "%s.%d.0" % (f, k), t, 2, 1), train)
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_branches(S, x, train, prefix=""):
    """stem, layer1, transitions and stages 2-4 (:462-501); returns the four branch outputs"""
    P = prefix
    x = F.relu(bn(S, P + "bn1", conv(S, P + "conv1", x, 2, 1), train))
    x = tap("stem", F.relu(bn(S, P + "bn2", conv(S, P + "conv2", x, 2, 1), train)))
    for b in range(_count(S, P + "layer1")):
        x = bottleneck(S, "%slayer1.%d" % (P, b), x, train)
    ys = [tap("layer1", x)]
    for si in (2, 3, 4):
        tp = "%stransition%d" % (P, si - 1)
        xs = []
        for i in range(si):
            src = ys[i] if i < len(ys) else ys[-1]
            if (tp + ".%d.0.weight" % i) in S:                  # Sequential(conv3x3, bn, relu)
                src = F.relu(bn(S, tp + ".%d.1" % i, conv(S, tp + ".%d.0" % i, src, 1, 1), train))
            elif (tp + ".%d.0.0.weight" % i) in S:              # Sequential(Sequential(conv3x3 s2, bn, relu), ...)
                for j in range(_count(S, tp + ".%d" % i)):
                    src = F.relu(bn(S, tp + ".%d.%d.1" % (i, j), conv(S, tp + ".%d.%d.0" % (i, j), src, 2, 1), train))
            xs.append(src)
        for m in range(_count(S, "%sstage%d" % (P, si))):
            xs = hr_module(S, "%sstage%d.%d" % (P, si, m), xs, train)
        ys = xs
        for i, y in enumerate(ys):
            tap("stage%d.b%d" % (si, i), y)
    return ys


def hrnet_concat(ys):
    size = ys[0].shape[-2:]
    return torch.cat([ys[0]] + [F.interpolate(y, size=size, mode="bilinear") for y in ys[1:]], 1)


def hrnetv2_forward(S, x, train=True):
    """HRNetv2.forward :462-514"""
    size = x.shape[-2:]
    f = hrnet_concat(hrnet_branches(S, x, train))
    y = F.relu(bn(S, "last_layer.1", conv(S, "last_layer.0", f), train))
    y = conv(S, "last_layer.3", y)
    return F.interpolate(y, size=size, mode="bilinear")
