"""Nearest resize / flip / mean-merge kernel and the TTA wrapper of BaseManager.infer against the CPU restatement."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.mark.parametrize("case", [((9, 13), (6, 9), 25), ((16, 24), (28, 42), 4), ((7, 5), (14, 10), 17), ((12, 20), (12, 20), 8),
                                  ((544, 960), (408, 720), 4), ((30, 41), (52, 71), 3)])
@pytest.mark.parametrize("compact", [False, True])
def test_resize_nearest_bit_exact(case, compact):
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd import ops
    (Hi, Wi), (Ho, Wo), C = case
    x = torch.randn(2, C, Hi, Wi, generator=torch.Generator().manual_seed(Hi))
    xd = ops.new_act(2, Hi, Wi, C, torch.device("cuda"), ld=C if compact else None, zero=True)   # compact: ld = C (e.g. 25)
    xd.copy_(x.permute(0, 2, 3, 1))
    ref = F.interpolate(x, size=(Ho, Wo), mode="nearest")
    assert torch.equal(ops.resize_nearest(xd, Ho, Wo).cpu().permute(0, 3, 1, 2), ref)
    assert torch.equal(ops.resize_nearest(xd, Ho, Wo, flip=1).cpu().permute(0, 3, 1, 2), F.interpolate(x.flip(3), size=(Ho, Wo), mode="nearest"))
    assert torch.equal(ops.resize_nearest(xd, Ho, Wo, flip=2).cpu().permute(0, 3, 1, 2), ref.flip(3))
    acc = ops.resize_nearest(xd, Ho, Wo)
    ops.resize_nearest(xd, Ho, Wo, flip=2, out=acc, accumulate=True, divide_by=2.0)
    assert torch.equal(acc.cpu().permute(0, 3, 1, 2), (ref + ref.flip(3)) / 2)


def test_tta_wrapper_matches_oracle(golden):
    """OCRNet-R50 (reference fixture weights), 10 augmented forward passes on each side"""
    _need_gpu()
    from oracle import nets as ON
    from oracle.state import fill_state
    from oracle.tta import tta_forward
    from miccai2021_cataract_semantic_segmentation_amd.models import OCRNet
    from miccai2021_cataract_semantic_segmentation_amd.utils.tta import SegmentationTTA
    g = golden("ocrnet_r50_e3_tiny")
    S = fill_state(json.loads(str(g["spec"])), int(g["seed"]))
    x = torch.from_numpy(g["x"])                                        # 2 x 3 x 64 x 96
    with torch.no_grad():
        want = tta_forward(lambda im: ON.ocrnet_forward(S, im, train=False)[1], x)
    model = OCRNet({"backbone": "resnet50", "out_stride": 8, "pretrained": False}, 3)
    model.load_state_dict(S)
    model.cuda().eval()
    model.get_intermediate = False
    got = SegmentationTTA(model)(x.cuda())
    assert got.shape == want.shape
    err = (got.cpu().double() - want.double()).abs().max().item()
    scale = want.abs().max().item()
    # ten eval forward passes through the folded-BN path: the eval-logit bar of test_nets_gpu (3e-3 of the logit scale)
    assert err <= 3e-3 * scale, (err, scale)
    assert (got.argmax(1).cpu() == want.argmax(1)).float().mean() > 0.995


def test_tta_in_manager_infer(tmp_path):
    """config['tta'] = True routes BaseManager.infer through the wrapper and leaves the manager's model untouched"""
    _need_gpu()
    from miccai2021_cataract_semantic_segmentation_amd.managers import DeepLabv3PlusManager, SyntheticCataractDataset
    cfg = {"name": "tta", "mode": "training", "manager": "DeepLabv3Plus", "log_path": str(tmp_path),
           "graph": {"model": "DeepLabv3Plus", "backbone": "resnet50", "out_stride": 16, "pretrained": False},
           "data": {"experiment": 2, "batch_size": 2},
           "loss": {"name": "LossWrapper", "losses": {"CrossEntropyLoss": 1}},
           "train": {"learning_rate": 1e-4, "epochs": 1}, "log_every_n_epochs": 1, "seed": 0}
    tr = SyntheticCataractDataset(4, 64, 96, 17, seed=1)
    va = SyntheticCataractDataset(2, 64, 96, 17, seed=2)
    m = DeepLabv3PlusManager(cfg, tr, va)
    m.train()
    plain = m.infer()
    m.config["tta"] = True
    tta = m.infer()
    assert len(tta) == 4 and all(np.isfinite(v) for v in tta) and all(np.isfinite(v) for v in plain)
    assert type(m.model).__name__ == "DeepLabv3Plus"
