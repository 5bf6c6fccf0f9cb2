"""Gather launches of csrc/pconv1.hip against the fp32 MFMA kernels on the strided / non-square layer shapes of the HRNet-W48 step at the bench size
(four tensors in turn): forward + BatchNorm partials, backward-data, backward-weight; microseconds per launch.   python3 tools/time_g1.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miccai2021_cataract_semantic_segmentation_amd import ops

dev = torch.device("cuda")
SHAPES = [(8, 136, 240, 48, 96, 3, 2, 1, "48->96 s2 @136x240"), (8, 136, 240, 48, 48, 3, 2, 1, "48->48 s2 @136x240"), (8, 68, 120, 96, 192, 3, 2, 1, "96->192 s2 @68x120"),
          (8, 68, 120, 48, 192, 3, 2, 1, "48->192 s2 @68x120"), (8, 34, 60, 192, 384, 3, 2, 1, "192->384 s2 @34x60"), (8, 136, 240, 256, 48, 3, 1, 1, "256->48 s1 @136x240"),
          (8, 136, 240, 256, 96, 3, 2, 1, "256->96 s2 @136x240"), (8, 272, 480, 64, 64, 3, 2, 1, "stem 64->64 s2 @272x480")]
NT = 4


def rec_of(t):
    r = ops.new_amax(dev)
    r[0] = t.abs().max().reshape(1).view(torch.int32)[0]
    return r


def timeit(fn, n=10):
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


print("%-26s %-6s %10s %10s %8s %9s" % ("layer", "op", "fp32 us", "gather us", "speedup", "TF-eq"))
for B, H, W, Cin, Cout, k, s, p, name in SHAPES:
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    xs = [torch.randn(B, H, W, Cin, device=dev).relu_() for _ in range(NT)]
    dys = [torch.randn(B, Ho, Wo, Cout, device=dev) * 1e-5 for _ in range(NT)]
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dw = torch.empty_like(w)
    outs = [torch.empty(B, Ho, Wo, Cout, device=dev) for _ in range(2)]
    dxs = [torch.empty(B, H, W, Cin, device=dev) for _ in range(2)]
    fl = 2.0 * B * Ho * Wo * Cout * Cin * k * k
    res = {}
    for mode in ("fp32", "g1"):
        for t in xs + dys:
            t._amax = rec_of(t) if mode == "g1" else None
        ops.release_b3_cache()
        res[mode, "fwd"] = timeit(lambda i: ops.conv_fwd(xs[i % NT], w, None, Cout, k, k, s, p, 1, out=outs[i % 2], bn_stats=True))
        res[mode, "dgrad"] = timeit(lambda i: ops.conv_bwd_data(dys[i % NT], w, (B, H, W, Cin), k, k, s, p, 1, out=dxs[i % 2]))
        res[mode, "wgrad"] = timeit(lambda i: ops.conv_bwd_weight(xs[i % NT], dys[i % NT], dw, None, k, k, s, p, 1))
    for op in ("fwd", "dgrad", "wgrad"):
        a, b = res["fp32", op], res["g1", op]
        print("%-26s %-6s %10.1f %10.1f %8.2f %9.1f" % (name, op, a, b, a / b, fl / b / 1e6))
