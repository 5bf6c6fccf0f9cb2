"""DeepLabv3+ on the HIP engine — constructor / forward contract and state-dict keys of the
reference's models/DeepLabv3Plus.py:10-175 (DeepLabv3Plus, ASPP :82-129, Decoder :132-175)."""
import torch
from torch import nn

from ..engine import (BatchNorm2d, Conv2d, EngineNet, bilinear, image_hw, concat_views, conv_bias, conv_bn_act, global_avgpool)
from ..utils import num_classes
from .backbone import ResNetBackbone, load_pretrained_trunk


def _bn(c):
    # the reference passes momentum=0.0003 positionally into eps (DeepLabv3Plus.py:85,98-104): eps=3e-4, momentum=0.1
    return BatchNorm2d(c, 0.0003)


class ASPP(nn.Module):
    def __init__(self, c_in, c_aspp, mult=1):
        super().__init__()
        self._c_in, self._c_aspp = c_in, c_aspp
        self.global_pooling = nn.AdaptiveAvgPool2d(1)
        self.relu = nn.ReLU(inplace=True)
        self.aspp1 = Conv2d(c_in, c_aspp, 1, 1, bias=False)
        self.aspp2 = Conv2d(c_in, c_aspp, 3, 1, dilation=int(6 * mult), padding=int(6 * mult), bias=False)
        self.aspp3 = Conv2d(c_in, c_aspp, 3, 1, dilation=int(12 * mult), padding=int(12 * mult), bias=False)
        self.aspp4 = Conv2d(c_in, c_aspp, 3, 1, dilation=int(18 * mult), padding=int(18 * mult), bias=False)
        self.aspp5 = Conv2d(c_in, c_aspp, 1, 1, bias=False)
        self.aspp1_bn, self.aspp2_bn, self.aspp3_bn = _bn(c_aspp), _bn(c_aspp), _bn(c_aspp)
        self.aspp4_bn, self.aspp5_bn = _bn(c_aspp), _bn(c_aspp)
        self.conv2 = Conv2d(c_aspp * 5, c_aspp, 1, 1, bias=False)
        self.bn2 = _bn(c_aspp)

    def run(self, cx, x):
        B, H, W, _ = x.shape
        c = self._c_aspp
        cat = torch.empty((B, H, W, 5 * c), dtype=torch.float32, device=x.device)
        parts = []
        for i in range(4):
            conv, bn = getattr(self, "aspp%d" % (i + 1)), getattr(self, "aspp%d_bn" % (i + 1))
            parts.append((conv_bn_act(cx, x, conv, bn, out=cat[..., i * c:(i + 1) * c]), i * c, (i + 1) * c))
        g = conv_bn_act(cx, global_avgpool(cx, x), self.aspp5, self.aspp5_bn)
        parts.append((bilinear(cx, g, H, W, True, out=cat[..., 4 * c:5 * c]), 4 * c, 5 * c))
        concat_views(cx, cat, parts)
        return conv_bn_act(cx, cat, self.conv2, self.bn2)


class Decoder(nn.Module):
    def __init__(self, c_low, c_aspp, num_classes, c_low_reduced=48, c_3x3=256):
        super().__init__()
        self._c_low_reduced, self._c_aspp = c_low_reduced, c_aspp
        self.relu = nn.ReLU(inplace=True)
        self.conv_low = Conv2d(c_low, c_low_reduced, 1, 1, bias=False)
        self.conv_low_bn = _bn(c_low_reduced)
        self.conv_3x3_1 = Conv2d(c_aspp + c_low_reduced, c_3x3, 3, 1, padding=1, bias=False)
        self.conv_3x3_1_bn = _bn(c_3x3)
        self.conv_3x3_2 = Conv2d(c_3x3, c_3x3, 3, 1, padding=1, bias=False)
        self.conv_3x3_2_bn = _bn(c_3x3)
        self.conv_out = Conv2d(c_3x3, num_classes, 1, 1)

    def run(self, cx, low, a):
        B, H, W, _ = low.shape
        c0, c1 = self._c_low_reduced, self._c_low_reduced + self._c_aspp
        cat = torch.empty((B, H, W, c1), dtype=torch.float32, device=low.device)
        x1 = conv_bn_act(cx, low, self.conv_low, self.conv_low_bn, out=cat[..., :c0])
        x2 = bilinear(cx, a, H, W, True, out=cat[..., c0:c1])
        concat_views(cx, cat, [(x1, 0, c0), (x2, c0, c1)])
        y = conv_bn_act(cx, cat, self.conv_3x3_1, self.conv_3x3_1_bn)
        return conv_bn_act(cx, y, self.conv_3x3_2, self.conv_3x3_2_bn, head=self.conv_out)


class DeepLabv3Plus(EngineNet):
    eligible_backbones = ["resnet50", "resnet101"]

    def __init__(self, config, experiment):
        super().__init__()
        self.backbone_name = config["backbone"] if "backbone" in config else "resnet50"
        self.c_aspp = config["aspp"]["channels"] if "aspp" in config else 256
        self.out_stride = config["out_stride"] if "out_stride" in config else 16
        self.config = config
        assert self.out_stride in [8, 16, 32]
        striding = {8: [False, True, True], 16: [False, False, True], 32: [True, True, True]}[self.out_stride]
        assert self.backbone_name in self.eligible_backbones, "backbone must be in {}".format(self.eligible_backbones)
        self.num_classes = num_classes(experiment)
        self.backbone_cutoff = {"layer1": "low", "layer4": "high"}
        self.backbone = ResNetBackbone(self.backbone_name, striding, self.backbone_cutoff)
        if config.get("pretrained", True):          # models/DeepLabv3Plus.py:33 (default True)
            load_pretrained_trunk(self.backbone, self.backbone_name, config)
        self.high_level_channels = self.backbone.out_channels("layer4")
        self.low_level_channels = self.backbone.out_channels("layer1")
        mult = 1 if self.out_stride >= 16 else 2
        self.aspp = ASPP(self.high_level_channels, self.c_aspp, mult)
        self.decoder = Decoder(self.low_level_channels, self.c_aspp, self.num_classes)
        if "projector" in config:
            raise NotImplementedError("the contrastive projector is outside the accelerated path")
        self.projector_model = None

    def _body(self, cx, x):
        H, W = image_hw(x)
        f = self.backbone.run(cx, x)
        a = self.aspp.run(cx, f["high"])
        logits = self.decoder.run(cx, f["low"], a)
        return [bilinear(cx, logits, H, W, True)]


class DeepLabv3(EngineNet):
    """models/DeepLabv3.py:11-71 of the reference: backbone 'out' = layer4 -> the same ASPP -> 1x1 classifier ->
    bilinear (align_corners=True) to the input size."""
    eligible_backbones = ["resnet50", "resnet101"]

    def __init__(self, config, experiment):
        super().__init__()
        self.backbone_name = config["backbone"] if "backbone" in config else "resnet50"
        self.c_aspp = config["aspp"]["channels"] if "aspp" in config else 256
        self.out_stride = config["out_stride"] if "out_stride" in config else 16
        self.config = config
        assert self.out_stride in [8, 16, 32]
        striding = {8: [False, True, True], 16: [False, False, True], 32: [True, True, True]}[self.out_stride]
        assert self.backbone_name in self.eligible_backbones, "backbone must be in {}".format(self.eligible_backbones)
        self.num_classes = num_classes(experiment)
        self.backbone_cutoff = {"layer4": "out"}
        self.backbone = ResNetBackbone(self.backbone_name, striding, self.backbone_cutoff)
        if config.get("pretrained", True):          # models/DeepLabv3.py:34 (default True)
            load_pretrained_trunk(self.backbone, self.backbone_name, config)
        self.backbone_out_channels = self.backbone.out_channels("layer4")
        self.aspp = ASPP(self.backbone_out_channels, self.c_aspp, 1 if self.out_stride >= 16 else 2)
        self.conv_out = Conv2d(self.c_aspp, self.num_classes, 1, 1)
        if "projector" in config:
            raise NotImplementedError("the contrastive projector is outside the accelerated path")
        self.projector_model = None

    def _body(self, cx, x):
        H, W = image_hw(x)
        a = self.aspp.run(cx, self.backbone.run(cx, x)["out"])
        return [bilinear(cx, conv_bias(cx, a, self.conv_out), H, W, True)]
