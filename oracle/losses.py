"""Loss / metric restatements (oracle; test infrastructure only)."""
import numpy as np
import torch
import torch.nn.functional as F

NUM_CLASSES = {1: 8, 2: 17, 3: 25}          # utils/defaults.py:232-237 via models/OCR.py:41-42
IGNORE_LABEL = {1: -100, 2: 17, 3: 25}      # losses/LossWrapper.py:17-24, losses/TwoScaleLoss.py:21-23


def lovasz_grad(gt_sorted):
    """losses/LovaszSoftmax.py:83-95"""
    gts = gt_sorted.sum()
    inter = gts - gt_sorted.cumsum(0)
    union = gts + (1 - gt_sorted).cumsum(0)
    jac = 1. - inter / union
    if len(gt_sorted) > 1:
        jac[1:] = jac[1:] - jac[:-1].clone()
    return jac


def lovasz_softmax(logits, target):
    """losses/LovaszSoftmax.py:19-61 with the defaults the shipped configs use
    (per_image=False, classes_to_ignore=None, classes_to_consider='present'):
    loops over the C logit channels, so label==C ("ignore") pixels are background
    for every class and are NOT masked out (:48-49 is a no-op)."""
    C = logits.shape[1]
    prob = F.softmax(logits, dim=1).permute(0, 2, 3, 1).reshape(-1, C)
    lbl = target.reshape(-1)
    losses = []
    for c in range(C):
        fg = (lbl == c).to(prob.dtype)
        if fg.sum() == 0:
            continue
        err = (fg - prob[:, c]).abs()
        err_sorted, perm = torch.sort(err, 0, descending=True)
        losses.append(torch.dot(err_sorted, lovasz_grad(fg[perm])))
    if not losses:
        return 0
    return sum(losses[1:], losses[0]) / len(losses) if len(losses) > 1 else losses[0]


def lovasz_softmax_np(logits, target):
    """float64 numpy cross-check of the same algorithm (loss value only; tie-invariant)."""
    x = np.asarray(logits, np.float64)
    C = x.shape[1]
    x = x - x.max(1, keepdims=True)
    p = np.exp(x)
    p /= p.sum(1, keepdims=True)
    p = np.moveaxis(p, 1, -1).reshape(-1, C)
    lbl = np.asarray(target).reshape(-1)
    tot, n = 0.0, 0
    for c in range(C):
        fg = (lbl == c).astype(np.float64)
        if fg.sum() == 0:
            continue
        err = np.abs(fg - p[:, c])
        o = np.argsort(-err, kind="stable")
        fs, es = fg[o], err[o]
        g = fs.sum()
        jac = 1.0 - (g - np.cumsum(fs)) / (g + np.cumsum(1 - fs))
        jac[1:] = jac[1:] - jac[:-1]
        tot += float(es @ jac)
        n += 1
    return tot / max(n, 1)


def resize_to_labels(logits, target):
    """the resize branch of the reference's losses (losses/TwoScaleLoss.py:45-48, losses/OhemCrossEntropy.py:23-26):
    F.upsample(input, size=(h, w), mode='bilinear') = F.interpolate(..., align_corners=False)"""
    if tuple(logits.shape[2:]) != tuple(target.shape[1:]):
        logits = F.interpolate(logits, size=tuple(target.shape[1:]), mode="bilinear", align_corners=False)
    return logits


def two_scale_lovasz(interm, final, target, w_interm=0.4, w_final=1.0):
    """losses/TwoScaleLoss.py:43-52"""
    return lovasz_softmax(final, target) * w_final + lovasz_softmax(resize_to_labels(interm, target), target) * w_interm


def cross_entropy(logits, target, experiment):
    """nn.CrossEntropyLoss(ignore_index=...) built at losses/LossWrapper.py:17-24"""
    return F.cross_entropy(logits, target, ignore_index=IGNORE_LABEL[experiment])


def ohem_cross_entropy(score, target, experiment=None, thresh=0.7, min_kept=100000):
    """losses/OhemCrossEntropy.py:22-39"""
    ignore = IGNORE_LABEL[experiment] if experiment in (2, 3) else -100
    score = resize_to_labels(score, target)
    pred = F.softmax(score, dim=1)
    pixel_losses = F.cross_entropy(score, target, ignore_index=ignore, reduction="none").reshape(-1)
    mask = target.reshape(-1) != ignore
    tmp = target.clone()
    tmp[tmp == ignore] = 0
    pred = pred.gather(1, tmp.unsqueeze(1)).reshape(-1)[mask]
    pred, ind = pred.sort()
    min_value = pred[min(max(1, min_kept), pred.numel() - 1)]
    threshold = max(float(min_value), thresh)
    pixel_losses = pixel_losses[mask][ind]
    return pixel_losses[pred < threshold].mean()


def confusion_matrix(logits, target):
    """utils/torch_utils.py:221-241: rows = prediction, cols = ground truth, int32;
    for C in {17, 25} label == C (ignore) is dropped."""
    C = logits.shape[1]
    pred = logits.transpose(1, 0).reshape(C, -1).argmax(0)
    t = target.reshape(-1).to(torch.int64)
    keep = t < C if C in (17, 25) else torch.ones_like(t, dtype=torch.bool)
    idx = pred[keep] * C + t[keep]
    return torch.bincount(idx, minlength=C * C).reshape(C, C).to(torch.int32)


CATEGORIES = {  # utils/defaults.py:16-33 (data, not code)
    1: dict(anatomies=[0, 4, 5, 6], instruments=[7], others=[1, 2, 3], rare=[2]),
    2: dict(anatomies=[0, 4, 5, 6], instruments=list(range(7, 17)), others=[1, 2, 3], rare=[16, 10, 9, 12, 14]),
    3: dict(anatomies=[0, 4, 5, 6], instruments=list(range(7, 25)), others=[1, 2, 3],
            rare=[24, 20, 21, 22, 18, 23, 19, 16, 12, 11, 14]),
}


def iou_per_class(cm, indices=None):
    """utils/torch_utils.py:307-332"""
    cm = cm.to(torch.float64)
    diag = cm.diag()
    den = cm.sum(0) + cm.sum(1) - diag
    iou = torch.where(den > 0, diag / den.clamp(min=1), torch.zeros_like(diag)).float()
    return iou if indices is None else iou[indices]


def mean_ious(cm, experiment):
    """t_get_mean_iou(categories=True, rare=True): (all, instruments, anatomies, rare)"""
    c = CATEGORIES[experiment]
    return tuple(float(iou_per_class(cm, i).mean()) for i in
                 (None, c["instruments"], c["anatomies"], c["rare"]))


def pixel_accuracy(cm):
    """utils/torch_utils.py:259-271"""
    d = cm.diag().float()
    rows = cm.sum(1).float()
    rows[rows == 0] = 1
    return float(d.sum() / cm.sum()), float((d / rows).mean())


def lr_multiplier(epoch, gamma=0.98):
    """utils/lr_functions.py:86-89 exponential with lr_params=None"""
    return gamma ** epoch


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as built at managers/BaseManager.py:441 (no wd, no amsgrad);
    in-place on p, m, v; ``step`` is the 1-based step count."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / (bc2 ** 0.5)).add_(eps), value=-lr / bc1)
