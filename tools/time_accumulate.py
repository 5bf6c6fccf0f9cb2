"""cost of accumulate=True (read-modify-write of the destination in the epilogue) in the backward-data kernels of the trunk"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, W, Ci, Co, k, s, p) in [(8, 136, 240, 48, 48, 3, 1, 1), (8, 68, 120, 96, 96, 3, 1, 1), (8, 34, 60, 192, 192, 3, 1, 1), (8, 17, 30, 384, 384, 3, 1, 1),
                                   (8, 136, 240, 256, 64, 1, 1, 0), (8, 136, 240, 48, 96, 3, 2, 1), (8, 68, 120, 96, 48, 1, 1, 0)]:
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Ho, Wo, Co, device=dev)
    dx = torch.zeros(B, H, W, Ci, device=dev)
    t0 = timeit(lambda: ops.conv_bwd_data(dy, w, (B, H, W, Ci), k, k, s, p, 1, out=dx, accumulate=False))
    t1 = timeit(lambda: ops.conv_bwd_data(dy, w, (B, H, W, Ci), k, k, s, p, 1, out=dx, accumulate=True))
    print("%dx%d %d->%d k%d s%d: write %.1f us, accumulate %.1f us" % (H, W, Ci, Co, k, s, t0, t1), flush=True)
