"""nn.CrossEntropyLoss(ignore_index=...) as the reference builds it (losses/LossWrapper.py:17-24,
losses/TwoScaleLoss.py:29-31), fused forward+backward on the HIP engine."""
import torch
from torch import nn

from .. import ops
from ._common import as_pixel_rows, scale_by


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, ignore_index, weight):
        rows = as_pixel_rows(pred.detach())
        lbl = target.reshape(-1)
        if lbl.dtype != torch.int64:
            lbl = lbl.long()
        dl = torch.empty_like(rows) if pred.requires_grad else None
        loss = ops.cross_entropy(rows, lbl.contiguous(), ignore_index, weight, dl)
        ctx.dl, ctx.shape = dl, pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dl = scale_by(ctx.dl, g)
        ctx.dl = None
        B, K, H, W = ctx.shape
        return dl.view(B, H, W, K).permute(0, 3, 1, 2), None, None, None


class CrossEntropyLoss(nn.Module):
    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, prediction, target):
        return _CEFn.apply(prediction, target, self.ignore_index, 1.0)
