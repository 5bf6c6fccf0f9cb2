"""Pointwise split-precision kernels (csrc/pconv1.hip) against the fp32 MFMA kernels on the 1 x 1 layer shapes of the HRNet-W48 / OCR step at the
bench size: forward (+ BatchNorm partials), backward-data, backward-weight; four tensors in turn (not cache resident); microseconds per launch,
TFLOP/s-equivalent, and the effective HBM rate (algorithmic bytes: operands read once, result written once).   python3 tools/time_p1.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops

dev = torch.device("cuda")
SHAPES = [(8 * 136 * 240, 64, 256, "layer1 64->256"), (8 * 136 * 240, 256, 64, "layer1 256->64"), (8 * 136 * 240, 64, 64, "layer1 64->64"),
          (8 * 136 * 240, 512, 256, "f_pixel.0"), (8 * 136 * 240, 256, 256, "f_pixel.3"), (8 * 136 * 240, 256, 512, "f_up"),
          (8 * 68 * 120, 96, 48, "fuse 96->48"), (8 * 34 * 60, 192, 96, "fuse 192->96"), (8 * 34 * 60, 192, 48, "fuse 192->48"),
          (8 * 17 * 30, 384, 192, "fuse 384->192")]
NT = 4


def rec_of(t):
    r = ops.new_amax(dev)
    r[0] = t.abs().max().reshape(1).view(torch.int32)[0]
    return r


def timeit(fn, n=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%-16s %-6s %10s %10s %8s %9s %9s" % ("layer", "op", "fp32 us", "p1 us", "speedup", "p1 TF-eq", "p1 TB/s"))
for rows, K, N, name in SHAPES:
    xs = [torch.randn(1, 1, rows, K, device=dev).relu_() for _ in range(NT)]
    dys = [torch.randn(1, 1, rows, N, device=dev) * 1e-5 for _ in range(NT)]
    w = (torch.randn(N, K, 1, 1, device=dev) * 0.1).contiguous(memory_format=torch.channels_last)
    dw = torch.empty_like(w)
    outs = [torch.empty(1, 1, rows, N, device=dev) for _ in range(2)]
    dxs = [torch.empty(1, 1, rows, K, device=dev) for _ in range(2)]
    fl = 2.0 * rows * K * N
    res = {}
    for mode in ("fp32", "p1"):
        for t in xs + dys:
            t._amax = rec_of(t) if mode == "p1" else None
        ops.release_b3_cache()
        res[mode, "fwd"] = timeit(lambda i: ops.conv_fwd(xs[i % NT], w, None, N, 1, 1, out=outs[i % 2], bn_stats=True))
        res[mode, "dgrad"] = timeit(lambda i: ops.conv_bwd_data(dys[i % NT], w, (1, 1, rows, K), 1, 1, out=dxs[i % 2]))
        res[mode, "wgrad"] = timeit(lambda i: ops.conv_bwd_weight(xs[i % NT], dys[i % NT], dw, None, 1, 1))
    by = {"fwd": 4.0 * rows * (K + N), "dgrad": 4.0 * rows * (K + N), "wgrad": 4.0 * rows * (K + N)}
    for op in ("fwd", "dgrad", "wgrad"):
        a, b = res["fp32", op], res["p1", op]
        print("%-16s %-6s %10.1f %10.1f %8.2f %9.1f %9.2f" % (name, op, a, b, a / b, fl / b / 1e6, by[op] / b / 1e6))
