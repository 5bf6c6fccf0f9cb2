"""TwoScaleLoss — losses/TwoScaleLoss.py:8-52 of the reference: w_final * L(final) + w_interm * L(interm)."""
import torch
from torch import nn

from ..utils import IGNORE_LABEL
from ._common import upsample_to_labels
from .cross_entropy import CrossEntropyLoss
from .lovasz import LovaszSoftmax
from .ohem import OhemCrossEntropy

CONCURRENT = True    # run the intermediate-scale loss on a side stream, concurrently with the final-scale loss
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device)
    return _SIDE[key]


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
        # "upsample intermediate if not already upsampled" (losses/TwoScaleLoss.py:45-48; F.upsample's default align_corners=False)
        logits_interm = upsample_to_labels(logits_interm, target)
        if CONCURRENT and logits_final.is_cuda:
            # the two losses are independent: the intermediate one runs on a side stream (its sort / scan passes interleave with
            # the final loss's; every kernel of a loss call is ordered on the stream it was issued to, workspaces are per stream)
            main = torch.cuda.current_stream(logits_final.device)
            side = _side_stream(logits_final.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                loss_interm = self.loss_interm(logits_interm, target)
            loss_final = self.loss_final(logits_final, target)
            main.wait_stream(side)
            logits_interm.record_stream(side)
            target.record_stream(side)
        else:
            loss_final = self.loss_final(logits_final, target)
            loss_interm = self.loss_interm(logits_interm, target)
        return loss_final * self.w_final + loss_interm * self.w_interm
