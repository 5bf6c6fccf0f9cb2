"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/catseg.h declares, validates arguments before touching the GPU, and the Python host
layer mirrors the reference's plugin surface (names, constructor contract, state-dict keys)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    inc = os.path.join(ROOT, "include")
    hdr = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    return sorted(set(re.findall(r"\b(catseg_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from miccai2021_cataract_semantic_segmentation_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "libcatseg_hip.so does not export %s" % s
    assert sorted(_lib.EXPORTS) == syms
    assert _lib.lib.catseg_version() >= 1


def test_argument_validation_without_gpu():
    """bad descriptors are rejected on the host, with a message, before any launch"""
    from miccai2021_cataract_semantic_segmentation_amd import _lib
    d = _lib.ConvDesc(1, 8, 8, 6, 8, 8, 16, 1, 1, 1, 0, 1, 8, 16, 0)   # Cin = 6 is not a multiple of 4
    rc = _lib.lib.catseg_conv2d_fwd(ctypes.byref(d), 16, 16, 0, 16, 0, None)
    assert rc == 1 and b"multiple of 4" in _lib.lib.catseg_last_error()
    d = _lib.ConvDesc(1, 8, 8, 8, 9, 8, 16, 3, 3, 1, 1, 1, 8, 16, 0)   # Ho inconsistent with the geometry
    assert _lib.lib.catseg_conv2d_fwd(ctypes.byref(d), 16, 16, 0, 16, 0, None) == 1
    assert _lib.lib.catseg_lovasz_softmax(16, 16, 100, 200, 1.0, 16, 0, 0, 16, 1 << 30, None) == 1  # K > 64
    assert _lib.lib.catseg_lovasz_workspace(4177920, 25) > 4177920 * 25 * 16
    with pytest.raises(_lib.CatsegError):
        _lib.check(1)


def test_no_cpu_fallback():
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.losses import LovaszSoftmax
    m = OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LovaszSoftmax({"experiment": 1})(torch.zeros(1, 8, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_plugin_surface_and_checkpoint_keys(golden):
    import miccai2021_cataract_semantic_segmentation_amd as pkg
    from miccai2021_cataract_semantic_segmentation_amd import models, losses
    for name in ("OCRNet", "DeepLabv3Plus"):
        assert hasattr(models, name)
    for name in ("LovaszSoftmax", "TwoScaleLoss", "LossWrapper", "CrossEntropyLoss"):
        assert hasattr(losses, name)
    for fx, cls, exp, K in (("ocrnet_r50_e3_tiny", models.OCRNet, 3, 25), ("deeplab_r50_e2_tiny", models.DeepLabv3Plus, 2, 17)):
        spec = json.loads(str(golden(fx)["spec"]))
        m = cls({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, exp)
        sd = m.state_dict()
        assert [k for k, _ in spec] == list(sd.keys())
        assert all(tuple(s) == tuple(sd[k].shape) for k, s in spec)
        assert m.num_classes == K and m.out_stride == 8 and m.projector_model is None
    # same seed -> same initial weights as the oracle's torchvision-ResNet restatement
    from oracle import resnet_tv
    torch.manual_seed(0)
    r = resnet_tv.resnet50(replace_stride_with_dilation=[False, True, True])
    torch.manual_seed(0)
    b = models.backbone.ResNetBackbone("resnet50", [False, True, True], {"layer3": "low", "layer4": "high"})
    assert torch.equal(r.layer4[2].conv3.weight, b["layer4"][2].conv3.weight)
    assert torch.equal(r.layer2[0].downsample[0].weight, b["layer2"][0].downsample[0].weight)


def test_flat_parameter_views_roundtrip():
    from miccai2021_cataract_semantic_segmentation_amd.engine import FlatParams
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    m = OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 1)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = FlatParams(m).ensure()
    after = m.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    w = m.backbone["layer1"][0].conv2.weight
    assert w.shape == (64, 64, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()      # physical OHWI
    assert w.grad is not None and w.grad.permute(0, 2, 3, 1).is_contiguous()
    assert fp.flat.numel() >= sum(p.numel() for p in m.parameters())
    m.load_state_dict(before)
    assert not fp._stale()
