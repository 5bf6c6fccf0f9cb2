"""LovaszSoftmax — same constructor / call contract as losses/LovaszSoftmax.py:8-32 of the reference,
computed by the fused HIP pipeline (catseg_lovasz_softmax): softmax, per-class error sort, Jaccard
gradient and the logits gradient in one pass over all classes."""
import torch
from torch import nn

from .. import ops
from ..utils import NUM_CLASSES
from ._common import as_pixel_rows, grad_like, scale_by


class _LovaszFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        rows = as_pixel_rows(pred.detach())
        lbl = target.reshape(-1)
        if lbl.dtype != torch.int64:
            lbl = lbl.long()
        lbl = lbl.contiguous()
        need_grad = pred.requires_grad
        dl = torch.empty_like(rows) if need_grad else None
        loss = ops.lovasz_softmax(rows, lbl, weight, dl)
        ctx.dl = dl
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dl = scale_by(ctx.dl, g)
        ctx.dl = None
        B, K, H, W = ctx.shape
        return dl.view(B, H, W, K).permute(0, 3, 1, 2), None, None


def lovasz_softmax(pred, target, weight=1.0):
    return _LovaszFn.apply(pred, target, weight)


class LovaszSoftmax(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.experiment = config["experiment"]
        self.num_classes = NUM_CLASSES.get(self.experiment)
        self.per_image = config.get("per_image", False)
        self.classes_to_ignore = config.get("classes_to_ignore", None)
        self.classes_to_consider = config.get("classes_to_consider", "present")
        if self.per_image or self.classes_to_ignore is not None or self.classes_to_consider != "present":
            raise NotImplementedError("only the configuration the shipped configs use is accelerated "
                                      "(per_image=False, classes_to_ignore=None, classes_to_consider='present')")

    def forward(self, prediction, target):
        return lovasz_softmax(prediction, target, 1.0)
