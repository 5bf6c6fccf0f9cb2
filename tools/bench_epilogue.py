"""cost of the BatchNorm-statistics epilogue: forward convolution with and without bn_stats on the HRNet branch shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, C) in [(8, 136, 240, 48), (8, 68, 120, 96), (8, 34, 60, 192), (8, 17, 30, 384)]:
    x = torch.randn(B, H, W, C, device=dev)
    w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, C, 3, 3, 1, 1, 1)
    t0 = timeit(lambda: ops.conv_fwd(x, w, None, C, 3, 3, 1, 1, 1, out=y))
    t1 = timeit(lambda: ops.conv_fwd(x, w, None, C, 3, 3, 1, 1, 1, out=y, bn_stats=True))
    fl = 2.0 * B * H * W * C * C * 9
    print("3x3 %3d->%3d @%dx%d: plain %.1f us (%.0f TF), with BN partials %.1f us (%.0f TF)" % (C, C, H, W, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)
