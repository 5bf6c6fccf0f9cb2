"""OCRNet on the HIP engine — same constructor / forward contract and state-dict keys as the
reference's models/OCR.py:10-138 (OCRNet, SpatialGatherModule :146-170, ObjectAttentionBlock2D
:173-284, SpatialOCR_Module :287-321)."""
import torch
from torch import nn

from .. import engine, ops
from ..engine import (BatchNorm2d, Conv2d, EngineNet, bilinear, image_hw, concat_views, conv_bn_act,
                      object_attention_core, spatial_gather)
from ..utils import num_classes
from .backbone import ResNetBackbone, load_pretrained_trunk
from .HRNetv2 import build_hrnet_trunk, concat_branches, run_hrnet_trunk


class ObjectAttentionBlock2D(nn.Module):
    def __init__(self, in_channels, key_channels, scale=1):
        super().__init__()
        if scale != 1:
            raise NotImplementedError("scale > 1 is unused by the reference configs")
        self.scale, self.in_channels, self.key_channels = scale, in_channels, key_channels
        self.pool = nn.MaxPool2d(kernel_size=(scale, scale))
        self.relu = nn.ReLU(inplace=True)

        def cbr(cin, cout):
            return [Conv2d(cin, cout, 1, bias=False), BatchNorm2d(cout), self.relu]
        self.f_pixel = nn.Sequential(*cbr(in_channels, key_channels), *cbr(key_channels, key_channels))
        self.f_object = nn.Sequential(*cbr(in_channels, key_channels), *cbr(key_channels, key_channels))
        self.f_down = nn.Sequential(*cbr(in_channels, key_channels))
        self.f_up = nn.Sequential(*cbr(key_channels, in_channels))

    def run(self, cx, x, proxy, K, out):
        q = conv_bn_act(cx, conv_bn_act(cx, x, self.f_pixel[0], self.f_pixel[1]), self.f_pixel[3], self.f_pixel[4])
        k = conv_bn_act(cx, conv_bn_act(cx, proxy, self.f_object[0], self.f_object[1]), self.f_object[3], self.f_object[4])
        v = conv_bn_act(cx, proxy, self.f_down[0], self.f_down[1])
        engine.tap("proxy", proxy), engine.tap("q", q), engine.tap("k", k), engine.tap("v", v)
        ctx = engine.tap("ctx", object_attention_core(cx, q, k, v, K, self.key_channels))
        return engine.tap("context", conv_bn_act(cx, ctx, self.f_up[0], self.f_up[1], out=out))


class SpatialOCR_Module(nn.Module):
    def __init__(self, in_channels, key_channels, out_channels, scale=1, dropout=0.0):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        self.object_context_block = ObjectAttentionBlock2D(in_channels, key_channels, scale)
        self.conv_bn_dropout = nn.Sequential(Conv2d(2 * in_channels, out_channels, 1, padding=0, bias=False),
                                             BatchNorm2d(out_channels), self.relu, nn.Dropout2d(dropout))
        self.in_channels = in_channels

    def run(self, cx, cat, feats, proxy, K, head=None):
        """cat: [B,H,W,2C] buffer whose upper half already holds feats (torch.cat([context, feats], 1)); head: the classifier that is the only
        consumer of the module's output (then returned: its logits)"""
        C = self.in_channels
        context = self.object_context_block.run(cx, feats, proxy, K, out=cat[..., :C])
        concat_views(cx, cat, [(context, 0, C), (feats, C, 2 * C)])
        return conv_bn_act(cx, cat, self.conv_bn_dropout[0], self.conv_bn_dropout[1], head=head, z_tap="ocr_out")


class SpatialGatherModule(nn.Module):
    def __init__(self, cls_num=0, scale=1):
        super().__init__()
        self.cls_num, self.scale = cls_num, scale


class HRNetFeatures(nn.Module):
    """HRNetv2 trunk (models/HRNetv2.py blocks) as an OCRNet backbone: returns the stride-4 concat of the
    four branches (15*width channels).  The reference's OCRNet raises NotImplementedError for HRNet
    (models/OCR.py:68-69); this is the build-side assembly SURVEY.md 8a/A2 describes."""

    def __init__(self, width=48, stage1_width=64, modules=(1, 4, 3)):
        super().__init__()
        self.out_channels = int(sum(build_hrnet_trunk(self, width, stage1_width, tuple(modules))))

    def run(self, cx, x, h2_consumers=None):
        return concat_branches(cx, run_hrnet_trunk(self, cx, x), h2_consumers)


class OCRNet(EngineNet):
    # resnet18/34 are broken in the reference; 'hrnet48' / 'hrnet32' are build-side additions
    eligible_backbones = ["resnet50", "resnet101", "hrnet48", "hrnet32", "hrnet18"]

    def __init__(self, config, experiment):
        super().__init__()
        self.config = config
        self.backbone_name = config["backbone"] if "backbone" in config else "resnet101"
        assert self.backbone_name in self.eligible_backbones, "backbone must be in {}".format(self.eligible_backbones)
        self.out_stride = config["out_stride"] if "out_stride" in config else 8
        assert self.out_stride in [8, 16, 32]
        self.align_corners = True
        self.dropout = config["dropout"] if "dropout" in config else 0.0
        if self.dropout != 0.0:
            raise NotImplementedError("Dropout2d(p>0) is not used by the shipped configs")
        self.num_classes = num_classes(experiment)
        self.get_intermediate = True
        self.relu = nn.ReLU(inplace=True)
        if "resnet" in self.backbone_name:
            strides = {8: [False, True, True], 16: [False, False, True], 32: [False, False, False]}[self.out_stride]
            self.backbone_cutoff = {"layer3": "low", "layer4": "high"}
            self.backbone = ResNetBackbone(self.backbone_name, strides, self.backbone_cutoff)
            self.high_out_channels = self.backbone.out_channels("layer4")
            if config.get("pretrained", True):      # models/OCR.py:44 (default True)
                load_pretrained_trunk(self.backbone, self.backbone_name, config)
            self.low_level_channels = self.backbone.out_channels("layer3")
        else:
            w = int(self.backbone_name[5:])
            hcfg = config.get("hrnet", {})
            self.backbone = HRNetFeatures(hcfg.get("width", w), hcfg.get("stage1_width", 64),
                                          hcfg.get("modules", (1, 4, 3)))
            self.out_stride = 4
            self.high_out_channels = self.low_level_channels = self.backbone.out_channels
        self.conv_high_map = nn.Sequential(Conv2d(self.high_out_channels, 512, 3, 1, 1), BatchNorm2d(512), self.relu)
        self.interm_prediction_head = nn.Sequential(
            Conv2d(self.low_level_channels, 512, 3, 1, 1), BatchNorm2d(512), self.relu, nn.Dropout2d(self.dropout),
            Conv2d(512, self.num_classes, 1, 1, 0, bias=True))
        self.spatial_gather = SpatialGatherModule(self.num_classes)
        self.spatial_ocr_head = SpatialOCR_Module(512, 256, 512, 1, self.dropout)
        self.conv_out = Conv2d(512, self.num_classes, 1, 1, bias=True)
        if "projector" in config:
            raise NotImplementedError("the contrastive projector is outside the accelerated path")
        self.projector_model = None

    def _body(self, cx, x):
        H, W = image_hw(x)
        K = self.num_classes
        if isinstance(self.backbone, HRNetFeatures):
            # (the two 3 x 3 head convolutions below are the only readers of the trunk's output: models/HRNetv2.concat_branches)
            low = high = self.backbone.run(cx, x, h2_consumers=[self.interm_prediction_head[0], self.conv_high_map[0]])
        else:
            f = self.backbone.run(cx, x)
            low, high = f["low"], f["high"]
        hd = self.interm_prediction_head
        engine.tap("concat", low)
        interm = engine.tap("interm_lowres", conv_bn_act(cx, low, hd[0], hd[1], head=hd[4]))
        B, h, w, _ = high.shape
        cat = torch.empty((B, h, w, 1024), dtype=torch.float32, device=x.device)
        feats = conv_bn_act(cx, high, self.conv_high_map[0], self.conv_high_map[1], out=cat[..., 512:])
        engine.tap("feats", feats)
        proxy = spatial_gather(cx, feats, interm, K)
        logits = self.spatial_ocr_head.run(cx, cat, feats, proxy, K, head=self.conv_out)
        engine.tap("logits_lowres", logits)
        up = bilinear(cx, logits, H, W, True)
        if self.get_intermediate:
            return [bilinear(cx, interm, H, W, True), up]
        return [up]
