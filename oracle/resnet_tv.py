"""torchvision-compatible ResNet restatement (oracle; test infrastructure only).

PARITY UNPINNED: the reference calls ``torchvision.models.resnet{18,34,50,101}``
(``models/OCR.py:5-6,58-61``, ``models/DeepLabv3Plus.py:5-6,35-41``,
``models/ResNet.py:33``) and torchvision is a third-party dependency that is
neither vendored in the reference nor installed here (``environment.yml:10``
lists a bare, unpinned ``torchvision``; README says pytorch 1.7 => 0.8.x).
This file restates torchvision's published ResNet v1.5 architecture:

* stem  conv1 7x7/2 p3 (no bias) -> bn1 -> relu -> maxpool 3x3/2 p1
* BasicBlock (expansion 1): 3x3(stride) bn relu 3x3 bn (+id) relu
* Bottleneck (expansion 4): 1x1 bn relu, 3x3(stride, dilation) bn relu, 1x1 bn (+id) relu
  - the stride sits on the 3x3 ("v1.5")
* downsample = 1x1 conv(stride) + BN when stride != 1 or channels change
* ``replace_stride_with_dilation[i]``: for layer{2,3,4}, ``dilation *= stride;
  stride = 1``; the FIRST block of the layer uses the PREVIOUS dilation.
* init: conv kaiming_normal(fan_out, relu); BN weight 1 / bias 0.
* child names conv1,bn1,relu,maxpool,layer1..4,avgpool,fc and block children
  conv1,bn1,conv2,bn2,conv3,bn3,relu,downsample.{0,1} (checkpoint key contract).
"""
from collections import OrderedDict

import torch
from torch import nn


def _c3(cin, cout, stride=1, dilation=1):
    return nn.Conv2d(cin, cout, 3, stride, dilation, dilation, bias=False)


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        if dilation > 1:
            raise NotImplementedError("dilation > 1 not supported in BasicBlock")
        self.conv1 = _c3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _c3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        width = planes
        self.conv1 = _c1(inplanes, width)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _c3(width, width, stride, dilation)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _c1(width, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, replace_stride_with_dilation=None):
        super().__init__()
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2, replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], 2, replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], 2, replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_c1(self.inplanes, planes * block.expansion, stride),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, previous_dilation)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


_CFG = {"resnet18": (BasicBlock, [2, 2, 2, 2]), "resnet34": (BasicBlock, [3, 4, 6, 3]),
        "resnet50": (Bottleneck, [3, 4, 6, 3]), "resnet101": (Bottleneck, [3, 4, 23, 3])}


def _factory(name):
    def make(pretrained=False, progress=True, **kw):
        # no network / no weights in the container: 'pretrained' is accepted and ignored
        return ResNet(*_CFG[name], **kw)
    make.__name__ = name
    return make


resnet18, resnet34, resnet50, resnet101 = (_factory(n) for n in ("resnet18", "resnet34", "resnet50", "resnet101"))


class IntermediateLayerGetter(nn.ModuleDict):
    """Restatement of torchvision.models._utils.IntermediateLayerGetter: keeps the
    children of ``model`` up to the last requested layer (same names), and returns an
    OrderedDict {return_layers[name]: activation}."""

    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
