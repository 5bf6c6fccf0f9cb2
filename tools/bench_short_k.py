"""the 256 x 256 blocked-plane bf16x3 kernel in the short-reduction regime (K = 432 ... 1728: the HRNet branch widths as INPUT channels,
256 / 512 output columns) against the default path: how much of its large-layer rate survives 27 ... 108 K-steps per tile
(162 TFLOP/s-equivalent at K = 864, 138 at K = 432: a 96- or 48-column variant for the branch layers would not pay for its split pass)"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, Ci, Co) in [(8, 68, 120, 96, 256), (8, 136, 240, 48, 256), (8, 34, 60, 192, 256), (8, 68, 120, 96, 512), (8, 136, 240, 720, 512)]:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = torch.empty(B, H, W, Co, device=dev)
    if Ci % 16 == 0:
        xb, wb = ops.split3_blocked(x)[0], ops.split3_weight_blocked(w)
        t = timeit(lambda: ops.conv_fwd_b3_blocked(tuple(x.shape), xb, wb, None, Co, 3, 3, 1, 1, 1, out=y))
        ts = timeit(lambda: ops.split3_blocked(x))
    else:
        t = ts = float("nan")
    fl = 2.0 * B * H * W * Co * Ci * 9
    t32 = timeit(lambda: ops.conv_fwd(x, w, None, Co, 3, 3, 1, 1, 1, out=y))
    print("3x3 %d->%d @%dx%d: b3w<blocked> %.1f us (%.0f TF-eq), blocked split %.1f us, default path %.1f us (%.0f TF)" % (Ci, Co, H, W, t, fl / t / 1e6, ts, t32, fl / t32 / 1e6), flush=True)
