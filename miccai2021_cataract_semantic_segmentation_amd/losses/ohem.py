"""OhemCrossEntropy — losses/OhemCrossEntropy.py:8-39 of the reference (online hard example mining: mean CE over the
pixels whose target-class probability is below max(thresh, the min_kept-th smallest probability)), fused forward +
backward on the HIP engine.  The reference sorts every pixel's probability; here the order statistic comes from a
radix select on device (catseg_ohem_cross_entropy) and nothing is copied to the host."""
import torch
from torch import nn

from .. import ops
from ..utils import IGNORE_LABEL
from ._common import as_pixel_rows, scale_by, upsample_to_labels


class _OhemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, target, ignore_label, thresh, min_kept):
        rows = as_pixel_rows(score.detach())
        lbl = target.reshape(-1)
        if lbl.dtype != torch.int64:
            lbl = lbl.long()
        dl = torch.empty_like(rows) if score.requires_grad else None
        loss = ops.ohem_cross_entropy(rows, lbl.contiguous(), ignore_label, thresh, min_kept, 1.0, dl)
        ctx.dl, ctx.shape = dl, score.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dl = scale_by(ctx.dl, g)
        ctx.dl = None
        B, K, H, W = ctx.shape
        return dl.view(B, H, W, K).permute(0, 3, 1, 2), None, None, None, None


class OhemCrossEntropy(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.thresh = config["thresh"] if "thresh" in config else 0.7
        self.min_kept = max(1, config["min_kept"]) if "min_kept" in config else 100000
        if "experiment" in config:
            self.ignore_label = IGNORE_LABEL[config["experiment"]] if config["experiment"] in (2, 3) else -100
        else:
            self.ignore_label = -100

    def forward(self, score, target, **kwargs):
        score = upsample_to_labels(score, target)      # losses/OhemCrossEntropy.py:23-26 (F.upsample, align_corners=False)
        return _OhemFn.apply(score, target, self.ignore_label, float(self.thresh), int(self.min_kept))
