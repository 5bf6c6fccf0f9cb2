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
        # forward: loss + d loss / d prob (kept in the call's own workspace); the logits gradient is formed in backward(), already multiplied
        # by the upstream scalar (round 5: the separate pass that scaled a stored gradient -- 418 MB read + written per loss -- is gone)
        loss, ws = ops.lovasz_softmax_fwd(rows, lbl, weight, want_grad=need_grad)
        ctx.rows, ctx.ws, ctx.weight = (rows, ws, weight) if need_grad else (None, None, weight)
        ctx.shape = pred.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        up = g.reshape(1)
        if up.dtype != torch.float32 or not up.is_contiguous():
            up = up.float().contiguous()
        dl = ops.lovasz_softmax_bwd(ctx.rows, ctx.ws, up, ctx.weight)
        ctx.rows = ctx.ws = None
        B, K, H, W = ctx.shape
        return dl.view(B, H, W, K).permute(0, 3, 1, 2), None, None


def lovasz_softmax(pred, target, weight=1.0):
    return _LovaszFn.apply(pred, target, weight)


class LovaszSoftmax(nn.Module):
    """losses/LovaszSoftmax.py:8-80.  The configuration every shipped config uses (per_image=False, classes_to_ignore=None,
    classes_to_consider='present') is ONE fused kernel sequence.  The other options are built on the same kernel:
      * per_image: one kernel call per image (slices of the NHWC logits), mean over images (:27-29);
      * classes_to_ignore = v: pixels labelled v are removed before the sort (:72-80) -- a torch gather of the remaining rows;
      * classes_to_consider = 'all' or a list S (:44-55): labels outside S are remapped to the no-class label, so the kernel
        sums exactly the PRESENT classes of S; a class of S without a foreground pixel contributes max_p prob_c (what
        dot(sort(errors), lovasz_grad(zeros)) evaluates to), added with torch ops; the mean runs over all of S.
    (these option paths use a few torch pointwise / reduction kernels; the default path does not)"""

    def __init__(self, config):
        super().__init__()
        self.experiment = config["experiment"]
        self.num_classes = NUM_CLASSES.get(self.experiment)
        self.per_image = config.get("per_image", False)
        self.classes_to_ignore = config.get("classes_to_ignore", None)
        self.classes_to_consider = config.get("classes_to_consider", "present")

    def forward(self, prediction, target):
        if not self.per_image:
            return self._flat(prediction, target)
        losses = [self._flat(prediction[b:b + 1], target[b:b + 1]) for b in range(prediction.shape[0])]
        return sum(losses) / len(losses)

    def _flat(self, prediction, target):
        K = prediction.shape[1]
        if self.classes_to_ignore is not None:
            if not prediction.is_cuda:
                raise RuntimeError("HIP losses need device tensors (no CPU fallback)")
            valid = target.reshape(-1) != self.classes_to_ignore
            rows = prediction.permute(0, 2, 3, 1).reshape(-1, K)[valid]
            target = target.reshape(-1)[valid].reshape(1, -1, 1)
            if rows.shape[0] == 0:
                return prediction.sum() * 0.0
            prediction = rows.t().reshape(1, K, -1, 1)           # NCHW view of the kept pixel rows (no copy)
        if isinstance(self.classes_to_consider, str) and self.classes_to_consider == "present":
            return lovasz_softmax(prediction, target, 1.0)
        S = list(range(K)) if isinstance(self.classes_to_consider, str) else [int(c) for c in self.classes_to_consider]
        if self.experiment in (2, 3) and K in S:                # the 'ignore' class is never summed (:48-49)
            S.remove(K)
        sel = torch.zeros(K + 1, dtype=torch.bool, device=prediction.device)
        sel[S] = True
        lbl = target.reshape(-1).long().clamp(0, K)
        remapped = torch.where(sel[lbl], lbl, torch.full_like(lbl, K)).reshape(target.shape)
        present = torch.bincount(lbl, minlength=K + 1)[S] > 0
        n_present = present.sum().to(torch.float32)
        loss_present = lovasz_softmax(prediction, remapped, 1.0)                       # mean over the present classes of S
        probs = torch.softmax(prediction.permute(0, 2, 3, 1).reshape(-1, K), dim=1)[:, S]
        absent = (probs.max(dim=0).values * (~present)).sum()
        return (loss_present * n_present + absent) / float(len(S))
