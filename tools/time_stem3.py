"""the HRNet stem's first convolution (3 -> 64, 3 x 3 / 2) at the bench size: direct kernels (csrc/stem3.hip) against the implicit-GEMM route"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops
dev = torch.device("cuda")
B, H, W = 8, 544, 960
xs = [torch.randn(B, 3, H, W, device=dev) for _ in range(4)]
w = (torch.randn(64, 3, 3, 3, device=dev) * 0.2).contiguous(memory_format=torch.channels_last)
dys = [torch.randn(B, H // 2, W // 2, 64, device=dev) * 1e-3 for _ in range(4)]
dw = torch.empty_like(w)


def t(fn, n=10):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        for i in range(4): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * n) * 1e3


def old_fwd(i):
    x4 = ops.nchw3_to_nhwc4(xs[i])
    wk = ops.weight_pad_cin(w, 64, 9, 3, 4)
    return ops.conv_fwd(x4, wk, None, 64, 3, 3, 2, 1, 1, bn_stats=True, train=True, exact=True)


def old_wgrad(i):
    x4 = ops.nchw3_to_nhwc4(xs[i])
    wk = ops.weight_pad_cin(w, 64, 9, 3, 4)
    dpk = torch.empty_like(wk)
    ops.conv_bwd_weight(x4, dys[i], dpk, None, 3, 3, 2, 1, 1)


print("forward + BN partials: implicit GEMM (+ repack) %.1f us, direct %.1f us   |   backward-weight: implicit GEMM (+ repack) %.1f us, direct %.1f us"
      % (t(old_fwd), t(lambda i: ops.stem3_fwd(xs[i], w, None, bn_stats=True)), t(old_wgrad), t(lambda i: ops.stem3_bwd_weight(xs[i], dys[i], dw))))
