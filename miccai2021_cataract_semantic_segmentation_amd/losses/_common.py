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


class _UpsampleFn(torch.autograd.Function):
    """F.upsample(input, size, mode='bilinear') of NCHW logits (align_corners=False, the default the reference's losses rely on:
    losses/TwoScaleLoss.py:45-48, losses/OhemCrossEntropy.py:23-26) on the HIP bilinear kernels, differentiable"""

    @staticmethod
    def forward(ctx, pred, h, w):
        from .. import ops
        x = pred.detach().permute(0, 2, 3, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        ctx.shape = tuple(x.shape)
        return ops.bilinear_fwd(x, h, w, False).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        from .. import ops
        gn = g.permute(0, 2, 3, 1)
        if not gn.is_contiguous():
            gn = gn.contiguous()
        return ops.bilinear_bwd(gn, ctx.shape, False).permute(0, 3, 1, 2), None, None


def upsample_to_labels(pred, target):
    """logits at label resolution: unchanged if they already are, else the bilinear resize of the reference's losses"""
    h, w = target.shape[-2:]
    if tuple(pred.shape[2:]) == (h, w):
        return pred
    if not pred.is_cuda:
        raise RuntimeError("HIP losses need device tensors (no CPU fallback)")
    return _UpsampleFn.apply(pred.float(), int(h), int(w))
