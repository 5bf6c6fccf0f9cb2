"""TwoScaleLoss — losses/TwoScaleLoss.py:8-52 of the reference: w_final * L(final) + w_interm * L(interm)."""
from torch import nn

from ..utils import IGNORE_LABEL
from .cross_entropy import CrossEntropyLoss
from .lovasz import LovaszSoftmax
from .ohem import OhemCrossEntropy

_REGISTRY = {"LovaszSoftmax": LovaszSoftmax, "CrossEntropyLoss": CrossEntropyLoss, "OhemCrossEntropy": OhemCrossEntropy}


class TwoScaleLoss(nn.Module):
    def __init__(self, config):
        super().__init__()
        iname, fname = config["interm"]["name"], config["final"]["name"]
        self.w_interm = config["interm"].get("weight", 0.4)
        self.w_final = config["final"].get("weight", 1.0)
        self.ignore_label = -100
        if "experiment" in config:
            exp = config["experiment"]
            self.ignore_label = IGNORE_LABEL[exp] if exp in (2, 3) else -100
        config["interm"].update({"experiment": config["experiment"]})
        config["final"].update({"experiment": config["experiment"]})
        if iname == "CrossEntropyLoss" and fname == "CrossEntropyLoss":
            self.loss_interm = CrossEntropyLoss(ignore_index=self.ignore_label)
            self.loss_final = CrossEntropyLoss(ignore_index=self.ignore_label)
        elif iname == fname:
            self.loss_interm = _REGISTRY[iname](config["interm"])
            self.loss_final = _REGISTRY[fname](config["final"])
        else:
            raise NotImplementedError("different losses for interm {} and final {}".format(config["interm"], config["final"]))

    def forward(self, logits_interm, logits_final, target):
        if logits_interm.shape[2:] != target.shape[1:]:
            raise NotImplementedError("intermediate logits must already be at label resolution (OCRNet upsamples them)")
        loss_final = self.loss_final(logits_final, target)
        loss_interm = self.loss_interm(logits_interm, target)
        return loss_final * self.w_final + loss_interm * self.w_interm
