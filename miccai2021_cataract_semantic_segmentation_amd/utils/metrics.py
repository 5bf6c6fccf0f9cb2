"""Confusion matrix / mIoU / pixel accuracy with the reference's conventions
(utils/torch_utils.py:221-346): rows = prediction, columns = ground truth, int32; for 17/25-class tasks the
ignore label (== K) is dropped; IoU_c = diag / (row + col - diag) with 0/0 -> 0; category means from
utils/defaults.py:16-33.  The matrix itself is one HIP kernel (argmax + LDS histogram)."""
import torch

from .. import ops
from .classes import CATEGORIES


def t_get_confusion_matrix(prediction, target, existing_matrix=None, no_ignore_class=True):
    """prediction: NCHW logits (CUDA); target: N(1)HW integer labels"""
    with torch.no_grad():
        K = prediction.shape[1]
        rows = prediction.permute(0, 2, 3, 1)
        if not rows.is_contiguous():
            rows = rows.contiguous()
        lbl = target.reshape(-1)
        if lbl.dtype != torch.int64:
            lbl = lbl.long()
        if not (no_ignore_class and K in (17, 25)) and int(lbl.max()) >= K:
            raise RuntimeError("label >= num_classes (one_hot would raise in the reference)")
        cm = existing_matrix if existing_matrix is not None else torch.zeros((K, K), dtype=torch.int32, device=prediction.device)
        return ops.confusion_matrix(rows.reshape(-1, K), lbl.contiguous(), cm)


def t_get_pixel_accuracy(confusion_matrix):
    with torch.no_grad():
        correct = torch.diag(confusion_matrix).to(torch.float)
        acc = torch.sum(correct) / torch.sum(confusion_matrix)
        sums = torch.sum(confusion_matrix, dim=1, dtype=torch.float)
        sums[sums == 0] = 1
        return acc, torch.mean(correct / sums)


def t_get_miou(confusion_matrix, experiment, indices=None, calculate_mean=True):
    if indices is None:
        indices = list(range(confusion_matrix.shape[0]))
    with torch.no_grad():
        diag = confusion_matrix.diag()[indices].to(torch.float)
        row = torch.sum(confusion_matrix, dim=0, dtype=torch.float)[indices]
        col = torch.sum(confusion_matrix, dim=1, dtype=torch.float)[indices]
        iou = diag / (row + col - diag)
        iou[iou != iou] = 0
        return iou.mean() if calculate_mean else iou


def t_get_mean_iou(confusion_matrix, experiment, categories=False, single_class=None, calculate_mean=True, rare=False):
    assert experiment in [1, 2, 3]
    if single_class is not None:
        raise NotImplementedError("single-class IoU is not used by the managers on the accelerated path")
    if not categories:
        return t_get_miou(confusion_matrix, experiment, calculate_mean=calculate_mean)
    c = CATEGORIES[experiment]
    out = (t_get_miou(confusion_matrix, experiment, calculate_mean=calculate_mean),
           t_get_miou(confusion_matrix, experiment, c["instruments"], calculate_mean),
           t_get_miou(confusion_matrix, experiment, c["anatomies"], calculate_mean))
    if rare:
        out = out + (t_get_miou(confusion_matrix, experiment, c["rare"], calculate_mean),)
    return out
