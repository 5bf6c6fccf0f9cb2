"""Per-model managers (names as main.py:46 resolves them: config['manager'] + 'Manager') and a synthetic
CaDIS-like dataset for smoke runs and benchmarks."""
import torch
from torch.utils.data import Dataset

from ..losses import LossWrapper
from .base import BaseManager


class OCRNetManager(BaseManager):
    """managers/OCRNet_Manager.py:12-245: the model returns (interm, final); TwoScaleLoss takes both."""

    def forward_loss(self, img, lbl):
        interm, out = self.model(img.float())
        if isinstance(self.loss, LossWrapper):
            return self.loss(None, out, lbl.long(), interm_prediction=interm), out
        return self.loss(interm, out, lbl.long()), out

    def final_output(self, out):
        return out[1] if isinstance(out, tuple) else out


class DeepLabv3PlusManager(BaseManager):
    """managers/DeepLabv3Plus_Manager.py:65-237: single output; LossWrapper-style or plain two-argument losses."""

    def forward_loss(self, img, lbl):
        out = self.model(img.float())
        if isinstance(self.loss, LossWrapper):
            return self.loss(None, out, lbl.long()), out
        return self.loss(out, lbl.long()), out


class DeepLabv3Manager(DeepLabv3PlusManager):
    """managers/DeepLabv3_Manager.py: a twin of the DeepLabv3Plus manager bar the class name (configs/DeepLabv3_rf_lvsz.json)."""


class HRNetv2Manager(DeepLabv3PlusManager):
    """(build-side name: the reference has no HRNetv2 manager; its HRNetv2 is a single-output net like DeepLabv3+)"""


class FCNManager(DeepLabv3PlusManager):
    """managers/FCN_Manager.py:10-17 of the reference: image in, logits out; Adam with a CONSTANT learning rate, or an exponential decay
    by config['train']['lr_decay_gamma'] per epoch (the other managers' polynomial schedule is not used)."""

    def load_optimiser(self):
        from ..optim import FusedAdam
        tc = self.config["train"]
        self.optimiser = FusedAdam(self.model, lr=tc["learning_rate"], grad_scale=self.grad_scale)
        self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimiser, tc["lr_decay_gamma"]) if "lr_decay_gamma" in tc else None


class EncDecManager(BaseManager):
    """managers/EncDec_Manager.py:14-274: the model is built from the TOP-LEVEL 'encoder' / 'decoder' config entries
    (:16-21, configs/UPN_rf_lvsz.json has no 'graph'), the loss is always a LossWrapper (:23-29), and the step is
    ``deep_features, prediction = model(img); loss = LossWrapper(deep_features, prediction, lbl, epoch=epoch)`` (:158-185)."""

    def load_model(self):
        from .. import dist as D
        from ..models import EncDec
        self.model = EncDec(self.config, self.experiment).to(self.device)
        if self.world > 1:
            D.broadcast_parameters(self.model)
            self.grad_scale = D.attach(self.model)
        else:
            self.grad_scale = 1.0

    def load_loss(self):
        self.config["loss"]["experiment"] = self.experiment
        self.config["loss"]["device"] = str(self.device)
        self.loss = LossWrapper(self.config["loss"])

    def forward_loss(self, img, lbl):
        deep_features, prediction = self.model(img.float())
        return self.loss(deep_features, prediction, lbl.long(), epoch=self.epoch), prediction

    def final_output(self, out):
        return out[1] if isinstance(out, tuple) else out      # (get_features = False at inference: prediction only)


class SyntheticCataractDataset(Dataset):
    """Seeded synthetic frames with blob-structured label maps (piecewise-constant patches, a few classes
    absent, optional ignore label) — the shape of data SURVEY.md 8d asks the measurements to use."""

    def __init__(self, n, height, width, num_classes, ignore=True, seed=0, patch=16):
        self.n, self.h, self.w, self.k, self.ignore, self.seed, self.patch = n, height, width, num_classes, ignore, seed, patch

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        p = self.patch
        lbl = torch.randint(0, self.k + (1 if self.ignore else 0), ((self.h + p - 1) // p, (self.w + p - 1) // p), generator=g)
        lbl[lbl == 3] = 0
        lbl = lbl.repeat_interleave(p, 0).repeat_interleave(p, 1)[: self.h, : self.w].contiguous()
        # image correlated with the labels so that a few optimisation steps can reduce the loss
        img = torch.rand(3, self.h, self.w, generator=g) * 0.5
        img[0] += (lbl.float() / (self.k + 1)) * 0.5
        img[1] += ((lbl * 7) % (self.k + 1)).float() / (self.k + 1) * 0.5
        return img, lbl, {"index": i}
