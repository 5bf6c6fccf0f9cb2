"""Encoder-decoder segmentation nets on the HIP engine: the reference's EncDec (models/EncDec.py:7-53) with
the ResNet / ResNeXt encoder wrappers (models/ResNet.py:5-94, models/ResNeXt.py:5-60: torchvision trunks
returning the four stage outputs) and the UPerNet decoder (models/UPerNet.py:7-145).  State-dict keys as
in the reference: ``enc_model.layer1.0.conv1.weight``, ``dec_model.ppm_conv.0.0.weight`` ..."""
import torch
from torch import nn

from ..engine import (BatchNorm2d, Conv2d, EngineNet, adaptive_avgpool, add_n, bilinear, concat_views, conv_bias,
                      conv_bn_act, copy_into, maxpool)
from ..utils import num_classes
from .backbone import BasicBlock, Bottleneck, _c1, load_pretrained_trunk


class _GroupedBottleneck(nn.Module):
    """torchvision Bottleneck with groups / width_per_group (ResNeXt); forward-only on the HIP path"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.0)) * groups
        self.conv1 = _c1(inplanes, width)
        self.bn1 = BatchNorm2d(width)
        self.conv2 = Conv2d(width, width, 3, stride, 1, groups=groups, bias=False)
        self.bn2 = BatchNorm2d(width)
        self.conv3 = _c1(width, planes * 4)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def run(self, cx, x):
        o = conv_bn_act(cx, x, self.conv1, self.bn1)
        o = conv_bn_act(cx, o, self.conv2, self.bn2, private_in=True)
        idt = x if self.downsample is None else conv_bn_act(cx, x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(cx, o, self.conv3, self.bn3, relu=True, residual=idt)


_ENC = {  # name -> (block, layers, groups, width_per_group)
    "ResNet18": (BasicBlock, [2, 2, 2, 2], 1, 64), "ResNet34": (BasicBlock, [3, 4, 6, 3], 1, 64),
    "ResNet50": (Bottleneck, [3, 4, 6, 3], 1, 64), "ResNet101": (Bottleneck, [3, 4, 23, 3], 1, 64),
    "ResNeXt50": (_GroupedBottleneck, [3, 4, 6, 3], 32, 4), "ResNeXt101": (_GroupedBottleneck, [3, 4, 23, 3], 32, 8),
}


class Encoder(nn.Module):
    """conv1/bn1/relu/maxpool/layer1..4 of a torchvision ResNet(-XT); run() returns the 4 stage outputs"""

    def __init__(self, name, config=None):
        super().__init__()
        block, layers, groups, wpg = _ENC[name]
        self.pretrained = (config or {}).get("pretrained", False)
        self._inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, 2, 3, bias=False)
        self.conv1.stem = True
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        for i, (planes, n) in enumerate(zip([64, 128, 256, 512], layers)):
            setattr(self, "layer%d" % (i + 1), self._make_layer(block, planes, n, 1 if i == 0 else 2, groups, wpg))
        nn.Linear(512 * block.expansion, 1000)  # torchvision's fc (RNG parity), dropped
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.pretrained:                 # models/ResNet.py:32-33: torchvision's ImageNet checkpoint
            load_pretrained_trunk(self, name, config)

    def _make_layer(self, block, planes, blocks, stride, groups, wpg):
        downsample = None
        if stride != 1 or self._inplanes != planes * block.expansion:
            downsample = nn.Sequential(_c1(self._inplanes, planes * block.expansion, stride), BatchNorm2d(planes * block.expansion))
        kw = dict(groups=groups, base_width=wpg) if block is _GroupedBottleneck else {}
        layers = [block(self._inplanes, planes, stride, downsample, **kw)]
        self._inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self._inplanes, planes, **kw))
        return nn.Sequential(*layers)

    def out_channels(self):
        outs = []
        for i in range(1, 5):
            blk = getattr(self, "layer%d" % i)[-1]
            outs.append((blk.conv3 if hasattr(blk, "conv3") else blk.conv2).out_channels)
        return outs

    def run(self, cx, x):
        x = conv_bn_act(cx, x, self.conv1, self.bn1, need_dx=False)
        x = maxpool(cx, x)
        outs = []
        for i in range(1, 5):
            for blk in getattr(self, "layer%d" % i):
                x = blk.run(cx, x)
            outs.append(x)
        return outs


def _cbr3(cin, cout):
    return nn.Sequential(Conv2d(cin, cout, 3, 1, 1, bias=False), BatchNorm2d(cout), nn.ReLU(inplace=True))


class UPerNet(nn.Module):
    def __init__(self, config, experiment):
        super().__init__()
        self.num_classes = num_classes(experiment)
        self.pool_scales = config.get("pool_scales", [1, 2, 3, 6])
        self.in_channels = config["input_channels"]
        self.in_scales = config["input_scales"]
        self.ppm_num_ch = config.get("ppm_num_ch", 512)
        self.fpn_num_ch = config.get("fpn_num_ch", 512)
        self.fpn_num_lvl = min(max(config.get("fpn_num_lvl", len(self.in_scales)), 1), len(self.in_scales))
        self.interpolate_result_up = config.get("interpolate_result_up", True)
        self.ppm_pooling = nn.ModuleList([nn.AdaptiveAvgPool2d(s) for s in self.pool_scales])
        self.ppm_conv = nn.ModuleList([nn.Sequential(Conv2d(self.in_channels[-1], self.ppm_num_ch, 1, bias=False),
                                                     BatchNorm2d(self.ppm_num_ch), nn.ReLU(inplace=True))
                                       for _ in self.pool_scales])
        self.ppm_last_conv = _cbr3(self.in_channels[-1] + len(self.pool_scales) * self.ppm_num_ch, self.fpn_num_ch)
        self.fpn_in = nn.ModuleList([nn.Sequential(Conv2d(c, self.fpn_num_ch, 1, bias=False), BatchNorm2d(self.fpn_num_ch),
                                                   nn.ReLU(inplace=True)) for c in self.in_channels[-self.fpn_num_lvl:-1]])
        self.fpn_out = nn.ModuleList([nn.Sequential(_cbr3(self.fpn_num_ch, self.fpn_num_ch)) for _ in range(self.fpn_num_lvl - 1)])
        self.conv_last = nn.Sequential(_cbr3(self.fpn_num_lvl * self.fpn_num_ch, self.fpn_num_ch),
                                       Conv2d(self.fpn_num_ch, self.num_classes, 1))

    def run(self, cx, conv_out):
        conv5 = conv_out[-1]
        B, h, w, c5 = conv5.shape
        npp, pc = len(self.pool_scales), self.ppm_num_ch
        cat = torch.empty((B, h, w, c5 + npp * pc), dtype=torch.float32, device=conv5.device)
        parts = [(copy_into(cx, conv5, cat[..., :c5]), 0, c5)]
        for k, (s, conv) in enumerate(zip(self.pool_scales, self.ppm_conv)):
            up = bilinear(cx, adaptive_avgpool(cx, conv5, s), h, w, False)
            c0 = c5 + k * pc
            parts.append((conv_bn_act(cx, up, conv[0], conv[1], out=cat[..., c0:c0 + pc]), c0, c0 + pc))
        concat_views(cx, cat, parts)
        feature = conv_bn_act(cx, cat, self.ppm_last_conv[0], self.ppm_last_conv[1])
        fpn = [feature]
        for i in range(2, self.fpn_num_lvl + 1):
            lat = self.fpn_in[-i + 1]
            conv_x = conv_bn_act(cx, conv_out[-i], lat[0], lat[1])
            feature = add_n(cx, [conv_x, bilinear(cx, feature, conv_x.shape[1], conv_x.shape[2], False)], relu=False)
            fo = self.fpn_out[-i + 1][0]
            fpn.append(conv_bn_act(cx, feature, fo[0], fo[1]))
        fpn.reverse()
        H, W = fpn[0].shape[1:3]
        fc = self.fpn_num_ch
        fus = torch.empty((B, H, W, self.fpn_num_lvl * fc), dtype=torch.float32, device=conv5.device)
        parts = [(copy_into(cx, fpn[0], fus[..., :fc]), 0, fc)]
        for i in range(2, self.fpn_num_lvl + 1):
            c0 = (i - 1) * fc
            parts.append((bilinear(cx, fpn[-i + 1], H, W, False, out=fus[..., c0:c0 + fc]), c0, c0 + fc))
        concat_views(cx, fus, parts)
        y = conv_bn_act(cx, fus, self.conv_last[0][0], self.conv_last[0][1])
        x = conv_bias(cx, y, self.conv_last[1])
        if self.interpolate_result_up:
            sc = self.in_scales[-self.fpn_num_lvl]
            x = bilinear(cx, x, H * sc, W * sc, False)
        return x


class EncDec(EngineNet):
    def __init__(self, config, experiment):
        super().__init__()
        self.config = config
        self.experiment = experiment
        self.enc_model = Encoder(config["encoder"]["model"], config["encoder"])
        if config["decoder"]["model"] != "UPerNet":
            raise NotImplementedError("only the UPerNet decoder is on the accelerated path")
        config["decoder"]["input_channels"] = self.enc_model.out_channels()
        config["decoder"]["input_scales"] = [4, 8, 16, 32]
        self.dec_model = UPerNet(config["decoder"], experiment)
        if "projector" in config:
            raise NotImplementedError("the contrastive projector is outside the accelerated path")
        self.projector_model = None
        self.get_features = True
        self.num_classes = self.dec_model.num_classes
        self.out_stride = 32

    def _body(self, cx, x):
        feats = self.enc_model.run(cx, x)
        pred = self.dec_model.run(cx, feats)
        return [feats[-1], pred] if self.get_features else [pred]
