"""standalone times of the main-stream (not overlapped) fp32 layers of the HRNet-W48 step against their HBM floor
(read x + read/write y once at 4.5 TB/s) and their fp32 MFMA floor (157 TFLOP/s)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = [  # B,H,W,Ci,Co,k,s,p   x count per step
    (8, 136, 240, 64, 256, 1, 1, 0, 5), (8, 136, 240, 256, 64, 1, 1, 0, 3), (8, 136, 240, 256, 512, 1, 1, 0, 1), (8, 136, 240, 512, 256, 1, 1, 0, 1),
    (8, 136, 240, 256, 256, 1, 1, 0, 1), (8, 136, 240, 256, 48, 3, 1, 1, 1), (8, 136, 240, 256, 96, 3, 2, 1, 1),
    (8, 272, 480, 64, 64, 3, 2, 1, 1), (8, 136, 240, 48, 96, 3, 2, 1, 8), (8, 136, 240, 48, 48, 3, 2, 1, 10), (8, 68, 120, 96, 192, 3, 2, 1, 8),
    (8, 68, 120, 96, 48, 1, 1, 0, 8), (8, 34, 60, 192, 48, 1, 1, 0, 7)]
tot = [0.0, 0.0]
for (B, H, W, Ci, Co, k, s, p, cnt) in SHAPES:
    x = torch.randn(B, H, W, Ci, device=dev)
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    y = ops.conv_fwd(x, w, None, Co, k, k, s, p, 1)
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    tf = timeit(lambda: ops.conv_fwd(x, w, None, Co, k, k, s, p, 1, out=y))
    td = timeit(lambda: ops.conv_bwd_data(dy, w, tuple(x.shape), k, k, s, p, 1, out=dx))
    tw = timeit(lambda: ops.conv_bwd_weight(x, dy, dw, None, k, k, s, p, 1))
    fl = 2.0 * y.numel() * Ci * k * k
    floor = max((x.numel() + y.numel()) * 4 / 4.5e12, fl / 157.3e12) * 1e6
    print("%dx%d %4d->%-4d k%d s%d x%-2d  fwd %6.1f  dgrad %6.1f  wgrad %6.1f us   floor %5.1f us  (%.1fx %.1fx %.1fx)"
          % (H, W, Ci, Co, k, s, cnt, tf, td, tw, floor, tf / floor, td / floor, tw / floor), flush=True)
    tot[0] += cnt * (tf + td + tw); tot[1] += cnt * 3 * floor
print("per step: %.2f ms measured, %.2f ms at the floors" % (tot[0] / 1e3, tot[1] / 1e3))
