"""Import shim for the REAL reference (build container only; never on the GPU box).

Used by ``tests/golden/make_golden.py`` to generate fixtures and by the optional
``-m refcheck`` tests.  It stubs the third-party modules the reference imports but
this image lacks (cv2, torchvision, tensorboard) *outside* the reference tree, plugs
the oracle's torchvision-ResNet restatement into the stubbed ``torchvision.models``,
and puts ``/root/reference`` on sys.path.  Nothing is written into /root/reference.
"""
import os
import sys
import types

REF = os.environ.get("CATSEG_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "models"))


def load():
    """Returns a namespace with the reference's ``models``, ``losses``, ``utils`` packages."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    sys.dont_write_bytecode = True
    import numpy as np
    import torch
    import torch.utils.data
    if not hasattr(np, "int"):
        np.int = int  # models/HRNetv2.py:359 uses the removed alias
    # utils/repeat_factor_sampling.py:71 calls Sampler.__init__(data_source=...)
    torch.utils.data.Sampler.__init__ = lambda self, data_source=None: None

    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(here))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import resnet_tv

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Dummy:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    def _nores(*a, **k):
        raise NotImplementedError("not available in the oracle harness")

    if "cv2" not in sys.modules:
        mod("cv2", INTER_NEAREST=0, INTER_LINEAR=1, BORDER_REFLECT_101=4, BORDER_CONSTANT=0,
            COLOR_BGR2RGB=4, imread=_nores, cvtColor=_nores, resize=_nores)
    tvm = mod("torchvision.models", resnet18=resnet_tv.resnet18, resnet34=resnet_tv.resnet34,
              resnet50=resnet_tv.resnet50, resnet101=resnet_tv.resnet101,
              resnext50_32x4d=_nores, resnext101_32x8d=_nores, wide_resnet50_2=_nores,
              wide_resnet101_2=_nores, inception_v3=_nores)
    tvu = mod("torchvision.models._utils", IntermediateLayerGetter=resnet_tv.IntermediateLayerGetter)
    tvm._utils = tvu
    tvt = mod("torchvision.transforms", **{n: _Dummy for n in (
        "ToPILImage", "ColorJitter", "ToTensor", "Normalize", "RandomApply", "Compose", "Grayscale")})
    mod("torchvision", models=tvm, transforms=tvt)
    tbw = mod("torch.utils.tensorboard.writer", SummaryWriter=_Dummy)
    mod("torch.utils.tensorboard", SummaryWriter=_Dummy, writer=tbw)

    if REF not in sys.path:
        sys.path.insert(0, REF)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import utils as ref_utils
        import losses as ref_losses
        import models as ref_models
    return types.SimpleNamespace(utils=ref_utils, losses=ref_losses, models=ref_models)
