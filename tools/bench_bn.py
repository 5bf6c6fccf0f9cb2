"""BatchNorm backward / apply on the HRNet-W48 activation shapes (bs 8 @ 544x960): event time per call and achieved GB/s.
usage: bench_bn.py [n]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
for (H, W, C) in [(136, 240, 48), (68, 120, 96), (34, 60, 192), (17, 30, 384), (136, 240, 256), (136, 240, 512)]:
    y = torch.randn(8, H, W, C, device=dev)
    dz = torch.randn_like(y)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    stats = torch.stack([y.mean((0, 1, 2)), 1.0 / (y.var((0, 1, 2), unbiased=False) + 1e-5).sqrt()]).contiguous()
    scale = (gamma * stats[1]).contiguous()
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    z = torch.empty_like(y)
    dy = torch.empty_like(y)
    def bwd(): ops.bn_backward(dz, None, y, stats, gamma, True, dg, db, dy_out=dy, beta=beta)
    mean = stats[0].contiguous()
    def app(): ops.bn_apply(y, mean, scale, beta, None, True, out=z)
    for name, fn, fl in (("backward", bwd, 5), ("apply", app, 2)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("bn %-8s 8x%dx%dx%-4d %6.1f MB  %7.1f us  %6.0f GB/s (algorithmic %d floats / element)" % (name, H, W, C, y.numel() * 4 / 1e6, ms * 1e3, fl * 4 * y.numel() / ms / 1e6, fl), flush=True)
