"""ResNet backbones on the HIP engine, key-compatible with torchvision's ResNet as the reference
uses it through IntermediateLayerGetter (models/OCR.py:58-61, models/DeepLabv3Plus.py:35-41):
children conv1, bn1, relu, maxpool, layer1..4; blocks conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}.
v1.5 bottleneck (stride on the 3x3); replace_stride_with_dilation as torchvision defines it."""
from collections import OrderedDict

import torch
from torch import nn

from ..engine import BatchNorm2d, Conv2d, conv_bn_act, maxpool


def _c3(cin, cout, stride=1, dilation=1):
    return Conv2d(cin, cout, 3, stride, dilation, dilation, bias=False)


def _c1(cin, cout, stride=1):
    return Conv2d(cin, cout, 1, stride, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _c1(inplanes, planes)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = _c3(planes, planes, stride, dilation)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = _c1(planes, planes * 4)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def run(self, cx, x):
        o = conv_bn_act(cx, x, self.conv1, self.bn1)
        o = conv_bn_act(cx, o, self.conv2, self.bn2, private_in=True)
        idt = x if self.downsample is None else conv_bn_act(cx, x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(cx, o, self.conv3, self.bn3, relu=True, residual=idt)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        if dilation > 1:
            raise NotImplementedError("dilation > 1 not supported in BasicBlock")
        self.conv1 = _c3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _c3(planes, planes)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def run(self, cx, x):
        o = conv_bn_act(cx, x, self.conv1, self.bn1)
        idt = x if self.downsample is None else conv_bn_act(cx, x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(cx, o, self.conv2, self.bn2, relu=True, residual=idt, private_in=True)


_CFG = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet34": (BasicBlock, [3, 4, 6, 3]),
        "resnet50": (Bottleneck, [3, 4, 6, 3]), "resnet101": (Bottleneck, [3, 4, 23, 3])}


class ResNetBackbone(nn.ModuleDict):
    """conv1..layer4 of a torchvision-style ResNet; ``run`` returns {out_name: NHWC activation}.
    Modules are created in torchvision's order (including a throw-away fc) so that a given RNG
    seed yields the same initial weights as the reference's ``resnet50(pretrained=False)``."""

    def __init__(self, name, replace_stride_with_dilation, return_layers):
        super().__init__()
        block, layers = _CFG[name]
        self._inplanes, self._dilation = 64, 1
        mods = OrderedDict()
        mods["conv1"] = Conv2d(3, 64, 7, 2, 3, bias=False)
        mods["conv1"].stem = True
        mods["bn1"] = BatchNorm2d(64)
        mods["relu"] = nn.ReLU(inplace=True)
        mods["maxpool"] = nn.MaxPool2d(3, 2, 1)
        rswd = list(replace_stride_with_dilation)
        mods["layer1"] = self._make_layer(block, 64, layers[0])
        mods["layer2"] = self._make_layer(block, 128, layers[1], 2, rswd[0])
        mods["layer3"] = self._make_layer(block, 256, layers[2], 2, rswd[1])
        mods["layer4"] = self._make_layer(block, 512, layers[3], 2, rswd[2])
        nn.Linear(512 * block.expansion, 1000)  # torchvision's fc: consumes the same RNG draws, then dropped
        for m in mods.values():
            for s in m.modules():
                if isinstance(s, nn.Conv2d):
                    nn.init.kaiming_normal_(s.weight, mode="fan_out", nonlinearity="relu")
                elif isinstance(s, nn.BatchNorm2d):
                    nn.init.constant_(s.weight, 1)
                    nn.init.constant_(s.bias, 0)
        last = max(i for i, k in enumerate(mods) if k in return_layers)
        for k, m in list(mods.items())[:last + 1]:
            self[k] = m
        self.return_layers = dict(return_layers)
        self.expansion = block.expansion

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        downsample = None
        prev = self._dilation
        if dilate:
            self._dilation *= stride
            stride = 1
        if stride != 1 or self._inplanes != planes * block.expansion:
            downsample = nn.Sequential(_c1(self._inplanes, planes * block.expansion, stride),
                                       BatchNorm2d(planes * block.expansion))
        layers = [block(self._inplanes, planes, stride, downsample, prev)]
        self._inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self._inplanes, planes, dilation=self._dilation))
        return nn.Sequential(*layers)

    def out_channels(self, layer):
        blk = self[layer][-1]
        return (blk.conv3 if hasattr(blk, "conv3") else blk.conv2).out_channels

    def run(self, cx, x_nchw):
        out = {}
        x = conv_bn_act(cx, x_nchw, self["conv1"], self["bn1"], need_dx=False)
        x = maxpool(cx, x)
        for name in ("layer1", "layer2", "layer3", "layer4"):
            if name not in self:
                break
            for blk in self[name]:
                x = blk.run(cx, x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("ResNetBackbone is executed by the owning EngineNet")


# ------------------------------------------------------------------------------------------------------------------
# ImageNet-pretrained trunks.  The reference calls ``torchvision.models.resnet50(pretrained=True)`` (models/OCR.py:44,59,
# DeepLabv3Plus.py:33-40, DeepLabv3.py:34-41, ResNet.py:32-33), which downloads torchvision's checkpoint.  torchvision and
# the network are not available to this package, so the SAME checkpoint file (torchvision state-dict format: keys
# ``conv1.weight``, ``layer1.0.conv1.weight`` ..., ``fc.*`` ignored) is read from disk:
#   config['pretrained_path'] (a file), or  $CATSEG_PRETRAINED_DIR/<arch>.pth  (e.g. resnet50.pth, resnext101_32x8d.pth).
# Without a file the trunk keeps its random initialisation and a loud warning says so (training from scratch gives a far
# lower mIoU than the reference's numbers).
TORCHVISION_FILES = {"resnet18": "resnet18", "resnet34": "resnet34", "resnet50": "resnet50", "resnet101": "resnet101",
                     "ResNet18": "resnet18", "ResNet34": "resnet34", "ResNet50": "resnet50", "ResNet101": "resnet101",
                     "ResNeXt50": "resnext50_32x4d", "ResNeXt101": "resnext101_32x8d"}


def load_pretrained_trunk(trunk, arch, config):
    """trunk: module whose state-dict keys are torchvision's (conv1, bn1, layer1 ...).  Returns the path loaded or None."""
    import os
    import warnings
    import torch
    path = (config or {}).get("pretrained_path")
    if not path and os.environ.get("CATSEG_PRETRAINED_DIR"):
        path = os.path.join(os.environ["CATSEG_PRETRAINED_DIR"], TORCHVISION_FILES.get(arch, arch) + ".pth")
    if not path or not os.path.isfile(path):
        warnings.warn("config asks for pretrained=True (%s) but no torchvision checkpoint was supplied (config['pretrained_path'] or "
                      "$CATSEG_PRETRAINED_DIR/%s.pth): the trunk is RANDOMLY INITIALISED -- results will not match the "
                      "reference's ImageNet-initialised runs" % (arch, TORCHVISION_FILES.get(arch, arch)), RuntimeWarning, stacklevel=2)
        return None
    sd = torch.load(path, map_location="cpu", weights_only=True)
    sd = sd.get("state_dict", sd)
    own = trunk.state_dict()
    take = {k: v for k, v in sd.items() if k in own and tuple(v.shape) == tuple(own[k].shape)}
    missing = [k for k in own if k not in take and not k.endswith("num_batches_tracked")]
    if missing:
        raise RuntimeError("pretrained checkpoint %s lacks %d trunk tensors (first: %s)" % (path, len(missing), missing[0]))
    trunk.load_state_dict(take, strict=False)
    return path
