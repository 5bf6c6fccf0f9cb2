"""LossWrapper — losses/LossWrapper.py:7-74 of the reference (weighted sum of named losses, exposing
loss_vals / total_loss / info_string), restricted to the losses on the accelerated path."""
import torch
from torch import nn

from ..utils import ce_ignore_index
from .cross_entropy import CrossEntropyLoss
from .lovasz import LovaszSoftmax
from .ohem import OhemCrossEntropy
from .two_scale import TwoScaleLoss


class LossWrapper(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.loss_weightings = config["losses"]
        self.device = config["device"]
        self.total_loss = None
        self.loss_classes, self.loss_vals = {}, {}
        names = []
        for name in self.loss_weightings:
            if name == "CrossEntropyLoss":
                fct = CrossEntropyLoss(ignore_index=ce_ignore_index(config["experiment"]))
            elif name == "LovaszSoftmax":
                fct = LovaszSoftmax(config)
            elif name == "TwoScaleLoss":
                fct = TwoScaleLoss(config)
            elif name == "OhemCrossEntropy":
                fct = OhemCrossEntropy(config)
            else:
                raise NotImplementedError("loss '{}' is outside the accelerated path".format(name))
            self.loss_classes[name] = fct
            self.loss_vals[name] = 0
            names.append(name)
        self.info_string = ", ".join(names)
        self.dc_off = "dc_off_at_epoch" in self.config

    def forward(self, deep_features, prediction, labels, loss_list=None, interm_prediction=None, epoch=None):
        self.total_loss = torch.tensor(0.0, dtype=torch.float, device=self.device)
        loss_list = list(self.loss_weightings.keys()) if loss_list is None else loss_list
        for name in self.loss_weightings:
            zero = torch.tensor(0.0, dtype=torch.float, device=self.device)
            if name not in loss_list:
                loss = zero
            elif name == "LovaszSoftmax":
                off = self.dc_off and epoch is not None and epoch < self.config["dc_off_at_epoch"]
                loss = zero if off else self.loss_classes[name](prediction, labels)
            elif name == "TwoScaleLoss":
                loss = self.loss_classes[name](interm_prediction, prediction, labels.long())
            else:
                loss = self.loss_classes[name](prediction, labels)
            loss = loss * self.loss_weightings[name]
            self.loss_vals[name] = loss
            self.total_loss = self.total_loss + loss
        return self.total_loss
