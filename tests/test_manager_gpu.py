"""Manager loop on the GPU: three tiny epochs, checkpoint keys, inference from the best checkpoint."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ocrnet_manager_train_validate_infer(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from miccai2021_cataract_semantic_segmentation_amd.managers import OCRNetManager, SyntheticCataractDataset
    cfg = {"name": "t", "mode": "training", "manager": "OCRNet", "log_path": str(tmp_path),
           "graph": {"model": "OCRNet", "backbone": "resnet50", "out_stride": 8, "pretrained": False},
           "data": {"experiment": 2, "batch_size": 2},
           "loss": {"name": "TwoScaleLoss", "interm": {"name": "LovaszSoftmax", "args": [], "weight": 0.4},
                    "final": {"name": "LovaszSoftmax", "args": [], "weight": 1.0}},
           "train": {"learning_rate": 1e-3, "epochs": 3}, "log_every_n_epochs": 1, "seed": 0}
    tr = SyntheticCataractDataset(8, 64, 96, 17, seed=1)
    va = SyntheticCataractDataset(3, 64, 96, 17, seed=2)
    m = OCRNetManager(cfg, tr, va)
    metrics = m.train()
    h = m.history
    assert len(h) == 3 and all(torch.isfinite(torch.tensor(r["train_loss"])) for r in h)
    assert h[-1]["train_loss"] < h[0]["train_loss"]                    # it learns
    assert abs(h[1]["lr"] - 1e-3 * 0.98 ** 2) < 1e-9                   # exponential 0.98 per epoch
    ck = torch.load(str(m.log_dir / "chkpts" / "chkpt_best.pt"), weights_only=False)
    assert set(ck) >= {"global_step", "epoch", "model_state_dict", "optimiser_state_dict", "best_loss", "best_miou", "is_best",
                       "scheduler_state_dict"}
    assert "backbone.layer3.0.conv1.weight" in ck["model_state_dict"] and ck["model_state_dict"]["conv_out.weight"].shape == (17, 512, 1, 1)
    assert (m.log_dir / "chkpts" / "chkpt_epoch_002.pt").exists() and (m.log_dir / "info.json").exists()
    # inference from the best checkpoint reproduces the best validation mIoU
    cfg2 = dict(cfg, mode="inference", load_checkpoint=m.run_id)
    inf = OCRNetManager(cfg2, None, va)
    miou = inf.infer()[0]
    assert abs(round(miou, 4) - metrics["best_miou"]) < 2e-3
