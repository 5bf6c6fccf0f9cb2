import torch


def as_pixel_rows(pred):
    """NCHW logits (ideally a channels_last view of the engine's NHWC output) -> contiguous [P, K]."""
    if pred.dim() != 4:
        raise ValueError("expected NCHW logits")
    if not pred.is_cuda:
        raise RuntimeError("HIP losses need device tensors (no CPU fallback)")
    nhwc = pred.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    return nhwc.reshape(-1, pred.shape[1])


def grad_like(pred, dl):
    B, K, H, W = pred.shape
    return dl.view(B, H, W, K).permute(0, 3, 1, 2)


def scale_by(dl, g):
    """dl *= g where g is a device scalar (the upstream gradient); no host sync"""
    from .. import ops
    return ops.scale_by_device_scalar(dl, g)
