"""Test-time augmentation of BaseManager.infer (managers/BaseManager.py:652-660 of the reference):

    ttach.SegmentationTTAWrapper(model, Compose([HorizontalFlip(), Scale([0.75, 1, 1.5, 1.75, 2])]), merge_mode='mean')

ttach is an un-pinned, un-vendored pip dependency of the reference (not installed here).  Restated from its published
behaviour (ttach 0.0.3): the Compose enumerates the product flip x scale (flip outer); an image is augmented flip-then-scale,
a mask de-augmented scale-back-then-flip-back; Scale resizes with F.interpolate(mode='nearest') to
(int(h * s), int(w * s)) and back with factor 1 / s (skipped for s == 1); the merger returns sum / n.
All resizing / flipping / averaging runs in one HIP kernel per augmentation (catseg_resize_nearest) on NHWC tensors."""
import torch

from ..engine import image_hw, is_nhwc4


class SegmentationTTA(torch.nn.Module):
    def __init__(self, model, scales=(0.75, 1, 1.5, 1.75, 2), flips=(False, True)):
        super().__init__()
        self.model, self.scales, self.flips = model, tuple(scales), tuple(flips)

    @torch.no_grad()
    def forward(self, image):
        from .. import ops
        x4 = image if is_nhwc4(image) else ops.nchw3_to_nhwc4(image.contiguous().float())
        H, W = image_hw(x4)
        n = len(self.flips) * len(self.scales)
        acc, i = None, 0
        for flip in self.flips:
            for s in self.scales:
                h2, w2 = (H, W) if s == 1 else (int(H * s), int(W * s))
                aug = x4 if (s == 1 and not flip) else ops.resize_nearest(x4, h2, w2, flip=1 if flip else 0)
                out = self.model(aug)
                if not torch.is_tensor(out):
                    raise RuntimeError("TTA needs a model that returns only the final logits (get_intermediate / get_features = False)")
                out = out.permute(0, 2, 3, 1)                       # NHWC view of the engine's output buffer
                hb, wb = (h2, w2) if s == 1 else (int(h2 * (1 / s)), int(w2 * (1 / s)))
                if (hb, wb) != (H, W):
                    raise RuntimeError("TTA scale %g does not round-trip %dx%d (-> %dx%d): the merge would mix shapes" % (s, H, W, hb, wb))
                i += 1
                if acc is None:
                    acc = ops.new_act(out.shape[0], H, W, out.shape[-1], out.device)
                ops.resize_nearest(out, H, W, flip=2 if flip else 0, out=acc, accumulate=i > 1, divide_by=float(n) if i == n else 0.0)
        return acc.permute(0, 3, 1, 2)
